/*
 * mp_api.h — C ABI of the MI355X spline-trajectory evaluation library (libmp_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of Geryyy/motion_planning: batched
 * evaluation of interpolating B-spline trajectories with the 15-term collision-aware
 * cost, its analytic (sub)gradient, the minimum clearance and the collision /
 * feasibility flags, plus the box-obstacle distance queries underneath it.
 *
 * The reference has no FFI; its seam is Python duck typing.  Every
 * entry point below names the reference interface (file:line under the reference
 * checkout) whose work it replaces.  Plain C: pointers and sizes only, no C++/torch
 * types.  Every function returns 0 on success or a negative MP_E_* code; the text of
 * the last error of a context is available from mp_last_error().
 *
 * Threading: a context is NOT re-entrant (callers serialise calls on one context);
 * different contexts are independent.  One context per GPU per process.
 * Current device: every entry point makes the context's GPU current for the duration
 * of the call and puts the caller's current HIP device back before it returns.
 */
#ifndef MP_API_H
#define MP_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MP_API_VERSION 2

/* ---- limits ------------------------------------------------------------------ */
#define MP_MAX_CTRL 16    /* waypoints per spline (start + vias + goal)            */
#define MP_MAX_SAMPLES 1024 /* n_samples_curve (the per-wave LDS slice of a gradient launch,
                               15 values + 4 bytes per sample, must fit the CU's 160 KB:
                               1024 fp64 samples are 125 KB, one wave per CU)     */
#define MP_NTERMS 15      /* cost terms, order of the weighted sum spline_opt.py:605-621 */

/* ---- error codes ------------------------------------------------------------- */
#define MP_OK 0
#define MP_E_ARG (-1)        /* bad argument (shape, range, NULL)                   */
#define MP_E_STATE (-2)      /* call order: scene/problem not set                   */
#define MP_E_HIP (-3)        /* a HIP runtime call failed (text in mp_last_error)   */
#define MP_E_UNSUPPORTED (-4)/* valid in the reference, not implemented here yet    */
#define MP_E_NOMEM (-5)

/* ---- enums ------------------------------------------------------------------- */
#define MP_F64 0
#define MP_F32 1

#define MP_MODEL_POINT 0 /* scene.signed_distance(p)            geom/scene.py:50-68  */
#define MP_MODEL_BOX 1   /* scene.signed_distance_block(...)    geom/scene.py:70-122 */

#define MP_MEM_HOST 0    /* X / outputs are host pointers (library stages them)    */
#define MP_MEM_DEVICE 1  /* X / outputs are device pointers on the context's GPU   */

/* index of each term in `terms[]` / `weights[]` (order of spline_opt.py:605-621)    */
enum mp_term {
    MP_T_LEN = 0,          /* path_length                      spline_opt.py:54-60   */
    MP_T_CURV = 1,         /* curvature_cost                   spline_opt.py:63-79   */
    MP_T_YAW_SMOOTH = 2,   /* yaw_smoothness_cost              spline_opt.py:100-106 */
    MP_T_SAFE = 3,         /* sum max(0,req-d)^2               spline_opt.py:546-547 */
    MP_T_SAFE_PREF = 4,    /* preferred clearance              spline_opt.py:548-556 */
    MP_T_REBOUND = 5,      /* approach rebound                 spline_opt.py:559-562 */
    MP_T_GOAL_CLEAR = 6,   /* goal clearance                   spline_opt.py:565-566 */
    MP_T_GOAL_TARGET = 7,  /* goal clearance target            spline_opt.py:567-569 */
    MP_T_APPROACH_CLEAR = 8, /* approach clearance             spline_opt.py:572-579 */
    MP_T_APPROACH_COL = 9, /* approach collision               spline_opt.py:580-581 */
    MP_T_VIA_DEV = 10,     /* via deviation                    spline_opt.py:584     */
    MP_T_YAW_DEV = 11,     /* yaw control deviation            spline_opt.py:585     */
    MP_T_YAW_MONO = 12,    /* yaw monotonicity                 spline_opt.py:588-593 */
    MP_T_YAW_SCHED = 13,   /* yaw schedule                     spline_opt.py:596-598 */
    MP_T_GOAL_NORMAL = 14  /* goal approach alignment          spline_opt.py:117-145 */
};

/* flags[] bits */
#define MP_FLAG_COLLISION 1u /* min_k d_k <  0                                      */
#define MP_FLAG_FEASIBLE 2u  /* min_k d_k >= required_clearance                     */

/*
 * mp_config — everything of one objective closure that does not depend on the
 * scenario: sizes, windows, weights, clearances.  It is the POD form of the
 * closure constants of objective_single (spline_opt.py:463-492, :500-641) and of the
 * 3-term variant (spline_opt.py:797-819).  All windows are integers derived ON THE
 * HOST with the reference's own floor/ceil expressions.
 *
 * Waypoint rows: W = [fixed head rows | free rows | fixed tail row].  For
 * optimize_bspline_path first_free = 1 (row 0 = start); for
 * optimize_bspline_with_vias first_free = 2 (rows 0,1 = start, via).  The optimised
 * vector is x = [free rows (n_free x 3, row-major) | free yaw controls (n_yaw_free)],
 * D = 3*n_free + n_yaw_free.
 */
typedef struct mp_config {
    int32_t n_samples;      /* S   n_samples_curve                                   */
    int32_t n_ctrl;         /* C   position waypoints, 4..MP_MAX_CTRL                */
    int32_t first_free;     /* index of the first optimised waypoint row             */
    int32_t n_free;         /* optimised waypoint rows (n_vias / n_additional_vias)  */
    int32_t n_yaw_ctrl;     /* Cy  yaw controls incl. the two fixed ends, >= 2       */
    int32_t n_yaw_free;     /* optimised yaw controls = Cy - 2                       */
    int32_t model;          /* MP_MODEL_POINT / MP_MODEL_BOX                         */
    int32_t n_safe;         /* S'  collision samples = len(idx_safe)                 */
    int32_t last_if_not_close; /* 1: idx_safe[S'-1] was appended conditionally and is
                                 dropped when allclose(P[idx_safe[S'-2]], P[S-1])
                                 (spline_opt.py:802-805); 0 otherwise                */
    int32_t keep_n;         /* preferred-clearance window d[:keep_n] (:550-554)      */
    int32_t use_pref;       /* pref > req and w_safe_preferred > 0 (:549)            */
    int32_t tail_start;     /* rebound window starts at d[tail_start] (:559-560)     */
    int32_t use_goal_target;/* goal_clearance_target given and weight > 0 (:568)     */
    int32_t goal_tail_n;    /* tail_n of _goal_approach_alignment_cost (:125)        */
    uint32_t term_mask;     /* bit t set: term t is part of the cost.  0x7fff for
                               optimize_bspline_path; len|curv|safe for the 3-term
                               objective (spline_opt.py:818)                         */
    int32_t reserved0;
    double weights[MP_NTERMS];
    double required_clearance;   /* :463 */
    double preferred_clearance;  /* :464-465 */
    double approach_target;      /* :577 */
    double goal_clearance_target;/* :569 */
    double point_radius;         /* geom/scene.py:50 (1e-6), point model only       */
    double half_extents[3];      /* moving box half sizes, box model only           */
    const int32_t *idx_safe;     /* [S'] sample index of each collision sample       */
    const uint8_t *approach_mask;/* [S'] 1 = sample is in the approach window (:572-576,
                                    fallback already applied by the host)            */
    const double *basis;         /* [S x C]  row-major sampling matrix M, or NULL to
                                    have the library build it (mp_build_basis)       */
    const double *basis_yaw;     /* [S x Cy] same for the yaw controls, or NULL      */
    const double *t_sched;       /* [S] clip(us / yaw_goal_reach_u, 0, 1) (:596)     */
} mp_config;

/*
 * mp_task — the scenario-dependent closure constants: one per scenario / restart
 * group.  Candidates of task t see boxes [box_begin, box_begin+box_count) of the
 * scene set with mp_set_scene.
 */
typedef struct mp_task {
    double ctrl[MP_MAX_CTRL][3]; /* waypoint rows: fixed rows hold start/via/goal,
                                    free rows hold via_init (:466, used by via_dev)  */
    double yaw_ctrl[MP_MAX_CTRL];/* yaw_ctrl_ref (:472): ends are start/goal yaw     */
    double goal_normal[3];       /* normalised resultant of the goal normals (:134-140) */
    int32_t has_goal_normal;     /* 0: term is 0 (:123, :141)                        */
    int32_t yaw_increasing;      /* goal_yaw >= start_yaw (:589)                     */
    int32_t box_begin;
    int32_t box_count;
} mp_task;

typedef struct mp_ctx mp_ctx; /* opaque */

/* ---- context ------------------------------------------------------------------ */

/* Create a context on GPU `device_id` computing in `dtype` (MP_F64 / MP_F32).
 * Fails with MP_E_HIP when no HIP device is available: there is no CPU fallback. */
int mp_create(int device_id, int dtype, mp_ctx **out);
int mp_destroy(mp_ctx *ctx);
const char *mp_last_error(const mp_ctx *ctx); /* ctx may be NULL: last create error */
int mp_api_version(void);

/* Use an externally owned hipStream_t (e.g. torch's current stream) for all work of
 * this context.  The handle is used as given: NULL is HIP's default (null) stream, which
 * is what torch reports as its default stream on ROCm.  mp_use_own_stream() goes back to
 * the non-blocking stream the context created for itself.  Switching waits for the work
 * queued on the previous stream, except when the new stream is being captured into a HIP
 * graph (no synchronisation is allowed then; ordering is the capturing caller's business).
 * MP_MEM_DEVICE calls of mp_eval can be captured: after one ordinary call of the same
 * shape they neither allocate nor synchronise. */
int mp_set_stream(mp_ctx *ctx, void *hip_stream);
int mp_use_own_stream(mp_ctx *ctx);
int mp_synchronize(mp_ctx *ctx);

/* HIP-event timer on the context's stream (bench.py times kernels with it). */
int mp_timer_start(mp_ctx *ctx);
int mp_timer_stop(mp_ctx *ctx, float *elapsed_ms); /* synchronises on the stop event */

/* ---- scene: replaces Scene.blocks / Block (geom/scene.py:10-45, geom/blocks.py:8-21)
 * SoA host arrays of box centres and HALF extents.  quat_xyzw may be NULL (all
 * identity); otherwise unit quaternions [x,y,z,w] (Block.quat, geom/blocks.py:8-21),
 * turned into rotations like the reference's quat_to_rot (geom/utils.py:3-14):
 *  - x = y = 0 (rotation about z): (cos, sin) = (1 - 2zz, 2wz); the pair is
 *    evaluated in the box's own frame with the axis-aligned formulas;
 *  - anything else (tilted box): the full matrix; the pair goes through the general
 *    OBB routine (15-axis SAT depth / closest vertex-face and edge-edge features),
 *    exact but ~30x the work of a z-aligned pair.
 * A quaternion that is not of unit length (1e-9) returns MP_E_ARG.
 * n_boxes == 0 is legal: distances are +inf (scene.py:112-113).  Replacing the scene
 * invalidates the current problem: call mp_set_problem again before mp_eval. */
int mp_set_scene(mp_ctx *ctx, int32_t n_boxes, const double *cx, const double *cy,
                 const double *cz, const double *hx, const double *hy, const double *hz,
                 const double *quat_xyzw /* [n_boxes x 4] or NULL */);

/* ---- problem: replaces the closure built by optimize_bspline_path
 * (geom/spline_opt.py:438-499) / optimize_bspline_with_vias (:776-796).
 * cfg and tasks are copied (including every array cfg points to). */
int mp_set_problem(mp_ctx *ctx, const mp_config *cfg, int32_t n_tasks, const mp_task *tasks);

/* Sampling matrix of an interpolating B-spline of degree k through n points at
 * u = linspace(0,1,n), evaluated at us = linspace(0,1,S):  P = M @ W.  Replaces the
 * per-call make_interp_spline of build_cubic_bspline / build_scalar_bspline
 * (geom/spline_opt.py:11-34).  Host-side, runs once per problem.  k in {1,2,3}. */
int mp_build_basis(int32_t n_ctrl, int32_t degree, int32_t n_samples, double *M /* [S x n] */);
/* Read back the matrices a context uses (tests feed them to the oracle). */
int mp_get_basis(mp_ctx *ctx, double *M /* [S x C] */, double *M_yaw /* [S x Cy] or NULL */);

/* ---- the hot path: replaces objective_single / objective_batch
 * (geom/spline_opt.py:500-641, :797-819, :301-302).
 *
 *   X         [n x D]      candidates, row-major, element type = context dtype
 *   task_of   [n] or NULL  task index per candidate; NULL: candidate i belongs to
 *                          task (i / cands_per_task) — see mp_set_cands_per_task —
 *                          (default: everything is task 0).  Host arrays are range-
 *                          checked (MP_E_ARG); a device array cannot be, so a candidate
 *                          whose entry lies outside [0, n_tasks) gets NaN cost / terms /
 *                          grad / min_clear and flags 0, and no table is read for it
 *   cost      [n]          weighted total j                         (required)
 *   terms     [n x 15]     unweighted terms, mp_term order          (or NULL)
 *   grad      [n x D]      d cost / d x (analytic sub-gradient)     (or NULL)
 *   min_clear [n]          min over collision samples of d          (or NULL)
 *   flags     [n]          MP_FLAG_* bits                           (or NULL)
 *   dist      [n x S']     clearance profile d_safe                 (or NULL)
 *
 * mem = MP_MEM_HOST: synchronous, results are in the host arrays on return.
 * mem = MP_MEM_DEVICE: asynchronous on the context's stream. */
int mp_eval(mp_ctx *ctx, int64_t n, const void *X, const int32_t *task_of, void *cost,
            void *terms, void *grad, void *min_clear, uint8_t *flags, void *dist, int mem);
int mp_set_cands_per_task(mp_ctx *ctx, int64_t cands_per_task);

/* ---- mp_eval for callers that cannot allocate ------------------------------------
 * No counterpart in the reference.  For callers that must not reach hipMalloc / hipFree:
 * code under HIP graph capture, tight serving loops, harnesses that forbid allocation.
 *
 * Same computation as mp_eval(..., MP_MEM_DEVICE) for the box model, always through the
 * split kernels (clearance_kernel writes the [n x S'] clearance / deciding-box tables,
 * eval_kernel reads them): the tables live in a file-scope __device__ array of
 * MP_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process
 * ends (its address is resolved once per device in mp_create).  The call allocates
 * nothing and frees nothing; results are bit-identical to mp_eval's.
 *
 *   X, task_of, cost, grad, min_clear, flags: DEVICE pointers, as in mp_eval (grad,
 *   min_clear, flags, task_of may be NULL); asynchronous on the context's stream.
 *   Limit: n * S' * (sizeof(T) + 4) (+ 255 bytes of alignment) <= MP_NOALLOC_SCRATCH_BYTES,
 *   i.e. fp64: 43,690 candidates at S' = 64, 10,922 at S' = 256; larger batches return
 *   MP_E_ARG with the sizes in mp_last_error.  Point model: MP_E_UNSUPPORTED (its
 *   kernels have no scratch; mp_eval never allocates for it).
 *   All calls on one GPU share that GPU's array, which keeps the last call's contents
 *   (nothing in it needs clearing: every entry a call reads it has written before).
 *   The library orders them: a call that comes on another stream than the previous one
 *   first waits on the device (hipStreamWaitEvent on an event recorded behind that call:
 *   no allocation, no host wait), so two contexts or two streams do not race on the
 *   tables.  The exception is a call made while its stream is being CAPTURED into a
 *   graph: it takes no part in that ordering (events of uncaptured work cannot enter a
 *   graph), and replays of the captured call that overlap other users of the array on
 *   the same GPU race on it; order them yourself.  Results are bit-identical to
 *   mp_eval's for every box-model launch whose mp_eval also runs the split kernels,
 *   and differ from a fused mp_eval launch only where the wide instance adds its sums
 *   in another order (S > 176: <= 1e-12 relative in cost and gradient; clearances and
 *   flags always the same bits). */
#define MP_NOALLOC_SCRATCH_BYTES (32u << 20)
int mp_eval_noalloc(mp_ctx *ctx, int64_t n, const void *X, const int32_t *task_of, void *cost,
                    void *grad, void *min_clear, uint8_t *flags);

/* ---- batched distance queries --------------------------------------------------
 * mp_point_sdf: Scene.signed_distance (geom/scene.py:50-68) for n points [n x 3].
 * mp_box_sdf:   Scene.signed_distance_block (geom/scene.py:70-122) for n poses of a
 *               yaw-rotated box: pos [n x 3], yaw_deg [n], one size (full extents).
 * Both use boxes [box_begin, box_begin+box_count) of the scene; element type =
 * context dtype; mem as in mp_eval. */
int mp_point_sdf(mp_ctx *ctx, int64_t n, const void *pts, double point_radius,
                 int32_t box_begin, int32_t box_count, void *out, int mem);
int mp_box_sdf(mp_ctx *ctx, int64_t n, const void *pos, const void *yaw_deg,
               const double size[3], int32_t box_begin, int32_t box_count, void *out,
               int mem);

/* ---- batched cross-entropy method, resident on the GPU --------------------------------
 * Replaces _cem_optimize (geom/spline_opt.py:222-277) and the objective_batch round trip per
 * generation (:301-304) for MANY independent instances (scenarios x restarts) in lock-step:
 * sampling, evaluation, elite refit and the stopping rule all run on the device; the host only
 * polls how many instances are still running.  Same update equations and stopping rule as the
 * reference; option names and defaults as in spline_opt.py:289-297. */
typedef struct mp_cem_options {
    int32_t population_size; /* 64 */
    int32_t max_iter;        /* 90 */
    double elite_frac;       /* 0.2  -> elite_count = max(2, ceil(pop * elite_frac)) */
    double tol;              /* 1e-3 */
    double alpha;            /* 0.7 */
    double min_sigma;        /* 1e-3 */
    uint64_t seed;           /* device Philox stream (ignored when `noise` is given) */
    uint64_t instance_base;  /* instance i of this call draws from stream instance_base + i: a job that shards a
                                list of instances over ranks passes the index of its first one, and the draws
                                (hence the results) do not depend on the partition; 0 for a single call   */
} mp_cem_options;

/*   task_of_inst [n_inst] or NULL  task of each instance (NULL: instance i -> task i / insts_per_task,
 *                                  insts_per_task = ceil(n_inst / n_tasks))
 *   x0, sigma0   [n_inst x D]      start mean / spread of every instance        (context dtype)
 *   noise        NULL, or [max_iter x n_inst x pop x D] standard-normal draws: generation g of instance i
 *                uses noise[g][i] instead of the device generator (seed-parity runs feed NumPy's stream)
 *   best_x [n_inst x D], best_val [n_inst], nit [n_inst]            results (best_x/best_val required)
 *   best_min_clear [n_inst], best_flags [n_inst]                    of the best candidate (or NULL)
 * All pointers are host (MP_MEM_HOST) or device (MP_MEM_DEVICE) pointers; synchronous on return. */
int mp_cem_solve(mp_ctx *ctx, int64_t n_inst, const int32_t *task_of_inst, const void *x0, const void *sigma0,
                 const mp_cem_options *opt, const void *noise, void *best_x, void *best_val, int32_t *nit,
                 void *best_min_clear, uint8_t *best_flags, int mem);

/* ---- the collective of a sharded job (SURVEY.md 8b `mp_allgather_best`, 8e) ---------------------------------
 * No counterpart in the reference (it is one process).  Candidates / scenarios are sharded over GPUs by
 * index, one process and one context per GPU; the only exchange of a job is one all-gather of every rank's results.
 * The Python host does it with torch.distributed (motion_planning_amd/batch.py); this is the same step for callers that
 * hold an RCCL communicator of their own:
 *
 *   nccl_comm       the caller's ncclComm_t (made with ncclCommInitRank on this context's GPU)
 *   send            DEVICE buffer of bytes_per_rank bytes: this rank's results, any layout (e.g. cost | min_clear | flags
 *                   of its candidates, written by mp_eval(MP_MEM_DEVICE) on this context)
 *   recv            DEVICE buffer of n_ranks * bytes_per_rank bytes: rank r's bytes at offset r * bytes_per_rank
 *
 * One ncclAllGather of raw bytes on the context's stream, ordered behind the kernels that wrote `send`; asynchronous
 * (mp_synchronize waits for it).  RCCL is not linked into the library: ncclAllGather is looked up in the calling process
 * (the caller created the communicator, so its librccl is loaded); MP_E_UNSUPPORTED when there is none. */
int mp_allgather_results(mp_ctx *ctx, void *nccl_comm, const void *send, void *recv, size_t bytes_per_rank);

#ifdef __cplusplus
}
#endif
#endif /* MP_API_H */
