// mp_kernels.hpp — the trajectory evaluation kernels (gfx950 / CDNA4).
//
// Mapping: ONE 64-lane wavefront per candidate trajectory.  The S spline samples live on
// the lanes (sample s on lane s % 64); the static boxes are wave-uniform and streamed
// through the scalar data path; all cross-sample stencils (np.gradient, second
// differences, tail windows) go through a per-wave LDS slice; per-trajectory sums use
// wavefront shuffles.  No MFMA: the spline product is a C<=16 term contraction per sample.
#pragma once
#include <type_traits>
#include "mp_device.hpp"

namespace mp {

#ifndef MP_F32_WAVES
#define MP_F32_WAVES 7   // fp32 kernels: 72 VGPRs, no scratch (measured at config 3, cost + gradient: 5 waves 0.331 ms, 6 0.310, 7 0.303, 8 0.348)
#endif
#ifndef MP_CLR_WAVES
#define MP_CLR_WAVES 6   // waves per SIMD clearance_kernel is register-budgeted for (measured: 4 0.934, 5 0.847, 6 0.792, 7 0.794, 8 0.815 ms)
#endif
#ifndef MP_WIDE_WAVES
#define MP_WIDE_WAVES 5   // waves per SIMD the wide split-mode instance (eval_kernel, NW = 3 | 4) is register-budgeted for (96 VGPRs:
                          // S=232, where five blocks fit the LDS, +5 % against the 4-wave budget; S=256, four blocks, unchanged)
#endif
#ifndef MP_COST64_WAVES
#define MP_COST64_WAVES 5   // waves per SIMD the fp64 cost-only kernels are register-budgeted for (<= 96 VGPRs)
#endif
#ifndef MP_GRAD64_WAVES
#define MP_GRAD64_WAVES 4   // waves per SIMD the fp64 gradient kernels are register-budgeted for
#endif
// Investigation builds only (tools/clearance_stats.py, tools/phase_clock.py build them into build_ab/):
//   -DMP_STATS   work counters of the clearance phase (contended atomics: timings of such a build mean nothing)
//   -DMP_PHASES  per-phase cycle clocks of eval_kernel, accumulated per wave and flushed once at kernel end
#ifdef MP_STATS
__device__ unsigned long long g_stats[8];
#define MP_STAT_ADD(i, v) do { if (lane == 0) atomicAdd(&g_stats[i], (unsigned long long)(v)); } while (0)
#else
#define MP_STAT_ADD(i, v) do { } while (0)
#endif
#ifdef MP_PHASES
__device__ unsigned long long g_phase[16];
#define MP_PHASE_DECL unsigned long long mp_ph[11] = {0}
#define MP_PHASE_INIT unsigned long long mp_t_prev = __builtin_readcyclecounter()
#define MP_PHASE(i) do { unsigned long long mp_t_now = __builtin_readcyclecounter(); \
        mp_ph[i] += mp_t_now - mp_t_prev; mp_t_prev = mp_t_now; } while (0)
#define MP_PHASE_FLUSH do { if (lane == 0) for (int mp_i = 0; mp_i < 11; ++mp_i) atomicAdd(&g_phase[mp_i], mp_ph[mp_i]); } while (0)
#else
#define MP_PHASE_DECL do { } while (0)
#define MP_PHASE_INIT do { } while (0)
#define MP_PHASE(i) do { } while (0)
#define MP_PHASE_FLUSH do { } while (0)
#endif
constexpr int kMaxCtrl = 16;
constexpr int kNTerms = 15;

template <typename T> struct DevTask {
    T ctrl[kMaxCtrl][3];
    T yaw_ctrl[kMaxCtrl];
    T normal[3];
    int has_normal, yaw_inc, box_begin, box_count;
    int sbox_begin;    // first record of this task's spatially sorted fp32 table (sboxes32)
    int spair_begin;   // first record of the same boxes (without the wide one) in the pair table, group by group
    int group_begin;   // first group of this task in the group table (records 2*group_begin ...)
    int group_count;   // groups of kGroupSize consecutive sorted boxes (1 group: no group test)
    int n_wide;        // 1: record 0 of the sorted table is a wide box handled before the sweep (clearance_chunk)
    // seed grid of the task's box range (clearance_chunk): cell (ix, iy, iz) of a uniform grid over the boxes' AABB holds the box
    // (index relative to the range) nearest to the cell's centre; grid_begin < 0: the range has none
    int grid_begin, gnx, gny, gnz;
    float gox, goy, goz, gix, giy, giz;   // origin and 1 / cell size per axis
    T xref[64];        // initial value of optimisation variable o (the via / yaw control it deviates from), o < D <= 56
};

template <typename T> struct EvalArgs {
    int S, C, ff, nf, Cy, nyf, Sp, last_cond, keep_n, use_pref, tail_start, use_goal_target, goal_tail_n;
    unsigned mask;
    int lds_per_wave;  // bytes
    int w_elems;       // elements of the waypoint table W in front of the records
    int five_waves;    // 1: the launch uses the five-waves-per-SIMD instance and its compact LDS slice (see eval_kernel, WV)
    int wide;          // > 1: the launch uses the wide instance (one candidate per block of `wide` waves; see eval_kernel, NW)
    int wide_rs;       // wide instance: values between the waves' shares in RED (a multiple of 8, >= the number of outputs D)
    int n_tasks;
    T w[kNTerms];
    T w2[kNTerms];     // 2 w[i] (exact): the factor of every d(e^2) in the reverse sweep
    T wf[kNTerms];     // w[i] where the term is in the objective (term_mask), 0 elsewhere; wf[1] = w_curv du.  What the launches
                       // without a term breakdown fold their partial sums with: one fma per term and no mask test.  (A term that
                       // is masked off keeps a zero or finite partial sum, so 0 * t adds nothing.)
    T req, pref, appr_tgt, goal_tgt, radius, ax, ay, az;
    TrigTab<T> trig;        // constants of yaw_rot (mp_device.hpp)
    T du, inv_du, inv_h2;   // 1/(S-1), 1/du, 1/(2 du) in T arithmetic (host: the same correctly rounded divisions)
    const int *idx_safe;
    const int *inv_idx;   // [S] collision sample of each spline sample, -1: none (read only when !idx_identity)
    int idx_identity;   // idx_safe[k] == k for every k (collision_check_subsample == 1): the table is not read
    const uint8_t *amask;
    const T *M;
    const T *My;
    const T *tsched;
    const DevTask<T> *tasks;
    const BoxRec<T> *boxes;
    const T *tilt;            // rotation matrices (9 values each) of the tilted boxes, indexed by their cb field; or NULL
    const BoxRec32 *boxes32;  // fp32 bound table, one record per box, sorted along a Morton curve per task; pad0 = original
                              // index (the kernels read the wide box, record 0, from it; the host builds pairs and groups from it)
    const BoxPair32 *pairs;   // the sorted boxes again, two per record (what the sweep reads)
    const BoxRec32 *groups;   // per group of consecutive sorted boxes two records: [2g] AABB, pad0/pad1 = first / count;
                              // [2g+1] centre of a representative box of the group
    float scene_radius;       // max |centre| + half extent over the scene (slack of the bound)
    const T *X;
    const int *task_of;
    long long cpt;
    // batched optimisers: candidates come in groups (one CEM population per group)
    int group_size;              // 0: not grouped
    const int *group_task;       // [n / group_size] task of each group, or NULL (task 0)
    const uint8_t *group_done;   // [n / group_size] 1: skip this group's candidates
    long long n;
    T *cost, *terms, *grad, *minc, *dist;
    uint8_t *flags;
    // split mode (large S): clearance_kernel fills these [n x Sp] tables, eval_kernel reads them
    T *pre_d;      // clearance of every collision sample
    int *pre_arg;  // its deciding box, relative to the task's range (-1: none)
    int defer;         // fused box-model launches with several 64-sample chunks: queue entries carry the trajectory's sample index
                       // (host: n_safe and the tasks' box counts fit the entry; clearance_chunk)
    const int *grid;   // seed grids of the tasks' box ranges, one int per cell (DevTask::grid_begin)
    const int *seed;   // cost-only launches of candidate groups (CEM populations): [n_groups x Sp] deciding box of every collision
                       // sample on the group's mean trajectory (nullptr: no seeds; set only together with group_size > 0)
};

// collision sample k -> spline sample (identity unless the path is subsampled for collision checks)
template <typename AT> __device__ __forceinline__ int sample_of(const AT &a, int k)
{
    return a.idx_identity ? k : a.idx_safe[k];
}

// One spline sample Q[s] = M[s,:] W (positions) and My[s,:] W[:,3] (yaw): the oracle's fixed fma chain over the
// waypoints.  Four waypoints (n_vias = 2, the reference's default everywhere) take a path of their own: the two rows
// are 32 bytes each and are fetched whole before the chain starts; the general loop waits for one entry at a time.
template <typename T> struct alignas(2 * sizeof(T)) Pair2 { T a, b; };

template <typename T>
__device__ __forceinline__ void spline_sample(const T *__restrict__ m, const T *__restrict__ my, int C, int Cy, const T *W,
                                              T &px, T &py, T &pz, T &yw)
{
    if (C == 4 && Cy == 4) {
        const Pair2<T> *r = reinterpret_cast<const Pair2<T> *>(m), *ry = reinterpret_cast<const Pair2<T> *>(my);
        const Pair2<T> r0 = r[0], r1 = r[1], y0 = ry[0], y1 = ry[1];
        px = r0.a * W[0]; py = r0.a * W[1]; pz = r0.a * W[2];
        px = mp_fma(r0.b, W[4], px); py = mp_fma(r0.b, W[5], py); pz = mp_fma(r0.b, W[6], pz);
        px = mp_fma(r1.a, W[8], px); py = mp_fma(r1.a, W[9], py); pz = mp_fma(r1.a, W[10], pz);
        px = mp_fma(r1.b, W[12], px); py = mp_fma(r1.b, W[13], py); pz = mp_fma(r1.b, W[14], pz);
        yw = y0.a * W[3];
        yw = mp_fma(y0.b, W[7], yw);
        yw = mp_fma(y1.a, W[11], yw);
        yw = mp_fma(y1.b, W[15], yw);
        return;
    }
    const T m0 = m[0];
    px = m0 * W[0]; py = m0 * W[1]; pz = m0 * W[2];
    for (int j = 1; j < C; ++j) {
        const T mj = m[j];
        px = mp_fma(mj, W[4 * j + 0], px);
        py = mp_fma(mj, W[4 * j + 1], py);
        pz = mp_fma(mj, W[4 * j + 2], pz);
    }
    yw = my[0] * W[3];
    for (int j = 1; j < Cy; ++j) yw = mp_fma(my[j], W[4 * j + 3], yw);
}

// The reference's default layout (n_vias = 2, combined_4d: four waypoints, the middle two free, two free yaw controls):
// the waypoints never go through LDS.  Fixed rows come from the task record (wave-uniform: scalar operands), the free
// ones are the candidate's x itself, read by every lane from the same 64 bytes.  Same fma chain, same bits.
// A wave-uniform value of a table that is read-only for the launch (task records, problem constants), read through the
// constant address space: it arrives over the scalar data path (s_load: a fraction of the latency of a vector load whose
// 64 lanes ask for the same address, no vector register) instead of global_load + v_readfirstlane.
template <typename U> __device__ __forceinline__ U ldu(const U *p)
{
    typedef const U __attribute__((address_space(4))) *CPtr;
    return *(CPtr)(unsigned long long)p;
}

template <typename T> struct TaskEnds { T s0, s1, s2, g0, g1, g2, y0, y3; };   // first / last waypoint, first / last yaw control
// read through the constant address space: the task table is read-only for the launch and the task is wave-uniform, so
// the eight values arrive over the scalar data path and stay scalar operands of the fma chain
template <typename T, typename TK> __device__ __forceinline__ TaskEnds<T> load_task_ends(const TK &tk)
{
    typedef const T __attribute__((address_space(4))) *CPtr;
    CPtr c = (CPtr)(unsigned long long)(&tk.ctrl[0][0]);
    CPtr y = (CPtr)(unsigned long long)(&tk.yaw_ctrl[0]);
    TaskEnds<T> e;
    e.s0 = c[0]; e.s1 = c[1]; e.s2 = c[2]; e.g0 = c[9]; e.g1 = c[10]; e.g2 = c[11];
    e.y0 = y[0]; e.y3 = y[3];
    return e;
}
template <typename T>
__device__ __forceinline__ void spline_sample4(const T *__restrict__ m, const T *__restrict__ my, const TaskEnds<T> &te, const T (&x)[8],
                                               T &px, T &py, T &pz, T &yw)
{
    const Pair2<T> *r = reinterpret_cast<const Pair2<T> *>(m), *ry = reinterpret_cast<const Pair2<T> *>(my);
    const Pair2<T> r0 = r[0], r1 = r[1], y0 = ry[0], y1 = ry[1];
    px = r0.a * te.s0; py = r0.a * te.s1; pz = r0.a * te.s2;
    px = mp_fma(r0.b, x[0], px); py = mp_fma(r0.b, x[1], py); pz = mp_fma(r0.b, x[2], pz);
    px = mp_fma(r1.a, x[3], px); py = mp_fma(r1.a, x[4], py); pz = mp_fma(r1.a, x[5], pz);
    px = mp_fma(r1.b, te.g0, px); py = mp_fma(r1.b, te.g1, py); pz = mp_fma(r1.b, te.g2, pz);
    yw = y0.a * te.y0;
    yw = mp_fma(y0.b, x[6], yw);
    yw = mp_fma(y1.a, x[7], yw);
    yw = mp_fma(y1.b, te.y3, yw);
}

// LDS slice of one wave (bytes).  Layout (S = samples, e = sizeof(T)):
//   W[w_elems] | REC[S][NREC]  | MINKEY[S] (u64; not in fp64 gradient launches) | ARG[S] (int) | QUEUE[128] (u32)
// REC is an array of per-sample RECORDS (field f of sample s at REC[s * NREC + f]), not one array per quantity: a
// lane addresses its sample once and reaches every field, and the fields of its neighbours s +- 1, through the
// immediate offset of the ds instruction.  (One array per quantity needs one base address per array; with 22 of
// them the scalar registers ran out and the compiler parked them in VGPR lanes, ~450 v_readlane per candidate.)
// NREC is odd, so that the 64 lanes of a ds_read_b64 / ds_read_b32 of one field fall into distinct banks.
//   fields of every launch:   P0 P1 P2 YW D1x D1y D1z                               ( 7, NREC =  7)
//   gradient launches add:    A1x A1y A1z A2x A2y A2z SEGW GD                       (15, NREC = 15)
// Fields with disjoint lifetimes share a slot:
//   D1* (first derivative) is dead once the curvature pass has run; the clearance phase then keeps CC, SS (cos / sin of
//       the collision sample's yaw) and D (its clearance) there; D in turn is dead after the clearance terms, and the
//       gather writes the yaw gradient GW into its slot;
//   A1* holds the adjoint of d1, then (in place) T1 = A1 + G^T A2; A2* is dead after that and takes the gather's GX GY GZ;
//   GD (d cost / d d_k) of an fp64 gradient launch is also where the sample's ordered clearance key lives during the
//       clearance phase (same 8 bytes, same owner lane), so those launches have no MINKEY array.
// The clearance-chain gradient is formed inside the gather loop (no arrays of its own).
// W holds 4 * max(C, Cy) elements, rounded up to 16 (EvalArgs::w_elems).
// fp64: S = 64 8.4 KB per wave with gradient, 4.9 KB without; S = 101 12.9 KB (12 waves per CU; 17 fields + MINKEY: 10).
enum RecField : int { kP0 = 0, kP1, kP2, kYW, kD1x, kD1y, kD1z,
                      kA1x, kA1y, kA1z, kA2x, kA2y, kA2z, kSEGW, kGD,
                      kCC = kD1x, kSS = kD1y, kD = kD1z, kGX = kA2x, kGY = kA2y, kGZ = kA2z, kGW = kD1z };
__host__ __device__ constexpr int rec_fields(bool grad) { return grad ? 15 : 7; }
// the ordered clearance keys live inside the records (field kGD) instead of an array of their own
__host__ __device__ constexpr bool key_in_record(bool grad, int e) { return grad && e == 8; }
__host__ __device__ inline int w_elems_for(int C, int Cy) { return ((4 * (C > Cy ? C : Cy)) + 15) & ~15; }
__host__ __device__ inline int lds_bytes_per_wave(int S, int e, bool grad, int w_elems)
{
    int b = (w_elems + rec_fields(grad) * S) * e;
    b = (b + 7) & ~7;
    if (!key_in_record(grad, e)) b += 8 * S;
    if (grad) b += 4 * S;   // deciding boxes: gradient launches only
    b += 4 * 128;
    return (b + 15) & ~15;
}
// slice of the five-wave instance (eval_kernel, WV = 5): records + 256 bytes (W / 16-bit queue / 16-bit deciding boxes)
__host__ __device__ inline int lds_bytes_per_wave_five(int S) { return (15 * 8 * S + 256 + 15) & ~15; }
// slice of the compact cost-only instance (WV = 6): records (7 fields) + keys + 256 bytes (W / 16-bit queue)
__host__ __device__ inline int lds_bytes_per_wave_compact_cost(int S) { return (7 * 8 * S + 8 * S + 256 + 15) & ~15; }
#ifndef MP_SEED_STRIDE
#define MP_SEED_STRIDE 8
#endif
// block of the wide instance (eval_kernel, NW > 1): records | deciding boxes | RED (the waves' shares of the sums; also the
// waypoint table during the spline pass); returns the offset of RED, `total` the bytes of dynamic LDS of the block
// (the deciding boxes start at the 8-byte aligned end of the records, exactly where eval_kernel puts `behind`: with fp32
// records and an odd S that is 4 bytes further than the records' own end, and RED must start behind THAT array)
__host__ __device__ inline int wide_red_offset(int S, int e) { return ((((rec_fields(true) * S * e + 7) & ~7) + 4 * S) + 15) & ~15; }
__host__ __device__ inline int wide_red_stride(int D) { return ((D > 2 ? D : 2) + 7) & ~7; }
__host__ __device__ inline int wide_block_bytes(int S, int e, int nw, int D, int w_elems)
{
    const int red = nw * wide_red_stride(D) > w_elems ? nw * wide_red_stride(D) : w_elems;
    return wide_red_offset(S, e) + ((red * e + 15) & ~15);
}
constexpr int kSeedStride = MP_SEED_STRIDE;   // the bound of a grouped scene is seeded from the representative of every 8th group
constexpr int kChunkRec = 7;   // record of clearance_kernel: P0 P1 P2 YW CC SS + 1 (odd)

// ---- culled clearance of one chunk of <= 64 collision samples (one per lane) against a task's boxes ----
// REC: the per-sample records (fields kP0 kP1 kP2 kYW read at the sample, kCC kSS written at the collision
// sample's own record; LOCAL: record = the lane's slot 0..63); MINKEY/ARG point at the chunk's first slot.  SEED: start from an upper bound taken from one representative box per group (used by the split
// kernel, which serves every scene large enough to have groups).  On return MINKEY[slot] holds the ordered key of the scene value, ARG[slot] the deciding box
// (-1: none) and the records' kCC/kSS the (cos, sin) of the sample's yaw.  KS: distance, in keys, between the keys of
// consecutive slots (1: an array; NREC: the keys are a field of the fp64 records).
// AS: the same for the deciding-box words (in ints).  Q16: work-queue entries are 16 bits (lane << 10 | box; every task < 1024 boxes).
// ARGS: keep the deciding box of every sample (the gradient needs it; cost-only launches skip the second atomic and its barriers).
template <typename T, int ROT, bool LOCAL, bool SEED, int NREC, int KS = 1, int AS = 1, bool Q16 = false, bool ARGS = true, bool SEEDED = false, typename AT, typename TK>
__device__ __forceinline__ void clearance_chunk(const AT &a, const TK &tk, const BoxRec<T> *boxes, const BoxRec32 *boxes32, const BoxPair32 *pairs,
                                                const BoxRec32 *groups, int ngroups, int n_wide, int lane, int k0, int Sp, T *REC,
                                                unsigned long long *MINKEY, int *ARG, unsigned int *QUEUE, const int *seed_row,
                                                int &qcarry, bool defer, bool keep_tail)
{
    // Slots: a LOCAL caller (clearance_kernel) hands in the 64 slots of this chunk, a fused caller the slots of the whole
    // trajectory; `own` is this lane's.  Queue entries are (sample << kshift | box); with `defer` (the same for every chunk
    // of a trajectory) the sample is the trajectory's collision sample index, so that entries of different chunks can share
    // an exact pass (`keep_tail`, below), without it the lane within the chunk.
    const int q0 = LOCAL ? 0 : k0;
    const int own = q0 + lane;
    const int kbase = defer ? 0 : q0;
    const int kshift = Q16 ? (defer ? 9 : 10) : (defer ? 22 : 24);
    const unsigned int kmask = (1u << kshift) - 1u;
    const float ax32 = (float)a.ax, ay32 = (float)a.ay, az32 = (float)a.az;
    // this lane's own collision sample
    const bool valid = k0 + lane < Sp;
    float px32 = 0.f, py32 = 0.f, pz32 = 0.f, exs = 0.f, eys = 0.f, ezs = 0.f, slack = 0.f;
    if (valid) {
        const int s = LOCAL ? lane : sample_of(a, k0 + lane);
        const T *rs = REC + s * NREC;
        T *rk = REC + (LOCAL ? lane : k0 + lane) * NREC;
        T c, sn;
        yaw_rot(rs[kYW], a.trig, c, sn);
        rk[kCC] = c; rk[kSS] = sn;
        MINKEY[own * KS] = key_init<T>();
        if (ARGS) ARG[own * AS] = -1;
        px32 = (float)rs[kP0]; py32 = (float)rs[kP1]; pz32 = (float)rs[kP2];
        const float c32 = __builtin_fabsf((float)c), s32 = __builtin_fabsf((float)sn);
        slack = 2e-6f * (__builtin_fabsf(px32) + __builtin_fabsf(py32) + __builtin_fabsf(pz32) +
                         a.scene_radius + ax32 + ay32 + az32) + 1e-30f;
        exs = __builtin_fmaf(s32, ay32, c32 * ax32) + slack;
        eys = __builtin_fmaf(c32, ay32, s32 * ax32) + slack;
        ezs = az32 + slack;
    }
    // the mover's inflated world AABB (the rounding of these six operations is far inside the slack).  Once the lane's
    // sample is known to penetrate some box by `depth`, the AABB is SHRUNK by a little less than that depth: the
    // penetration value of a pair is minus the smallest projection overlap over all directions, so it cannot exceed
    // the overlap of the two world AABBs along x, y or z; against the shrunk AABB a box whose overlap along some axis
    // is below the current depth shows a positive gap (lb2 > 0 = U32) and is culled by the very same test, while a
    // box that could be as deep or deeper still gives lb2 = 0.  Group AABBs contain their boxes, so the group test
    // inherits the bound.  (In a cluttered scene a penetrating sample overlaps the AABBs of ~20 boxes; one or two of
    // them can be the deepest.)
    float mlox = px32 - exs, mloy = py32 - eys, mloz = pz32 - ezs, mhix = px32 + exs, mhiy = py32 + eys, mhiz = pz32 + ezs;
    // running upper bound of the lane's squared clearance; -1 on lanes without a sample, so that `lb2 <= U32`
    // alone (one v_cmp, whose result is the ballot) says whether the lane needs a pair
    float U32 = valid ? __builtin_huge_valf() : -1.0f;
    const unsigned int lane_hi = (unsigned int)(defer ? own : lane) << kshift;
    // Work queue: a LINEAR array of up to 128 entries, filled from the front.  An exact pass takes entries 0 .. 63 and moves
    // what lies behind them (fewer than 64 entries) to the front, so a push needs no ring arithmetic: its slot is
    // qcount + (needing lanes below this one), which v_mbcnt delivers with qcount as its start value.
    int qcount = qcarry;  // wave-uniform; entries a previous chunk of the trajectory left behind (defer)
    MP_STAT_ADD(0, 1);
    // a seed box per sample (below): from the caller's row, else from the task's seed grid
    const int grid_begin = SEEDED ? ldu(&tk.grid_begin) : -1;
    const bool seeded = SEEDED && (seed_row || (grid_begin >= 0 && a.grid));
    if (SEED && ngroups > 1 && !seeded) {
        // Seed of the upper bound: the mover's centre lies in the mover and a box's centre in the box, so the scene
        // value is at most the distance from this sample to the centre of any box.  One representative box per
        // group (record 2g+1), coordinates inflated by the slack, keeps the first groups of the sweep from
        // being evaluated against an infinite bound.  Every kSeedStride-th group is enough: the groups are visited
        // nearest to the nominal path first and the bound tightens within the first exact passes (measured on the
        // config-4 shape: every group 1.013 ms, every 2nd 0.960, 4th 0.941, 8th 0.932, 16th .. none 0.940).
        float ub = __builtin_huge_valf();
        for (int g = 0; g < ngroups; g += kSeedStride) {
            const BoxRec32 rep = load_box32_uniform(groups, 2 * g + 1);
            const float dx = __builtin_fabsf(px32 - rep.lox) + slack, dy = __builtin_fabsf(py32 - rep.loy) + slack,
                        dz = __builtin_fabsf(pz32 - rep.loz) + slack;
            ub = __builtin_fminf(ub, __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
        }
        if (valid) U32 = ub * (1.0f + 4e-6f);
    }

    // exact evaluation of the first `cnt` <= 64 queued (sample, box) pairs, one per lane; `rest` entries behind them move
    // to the front of the queue
    auto drain = [&](int cnt, int rest) {
        MP_STAT_ADD(3, cnt); MP_STAT_ADD(4, 1);
        wave_lds_sync();
        const bool act = lane < cnt;
        unsigned long long key = ~0ull, oldk = 0ull;
        int kk = 0, bb = 0;
        unsigned int moved = 0;
        if (lane < rest) moved = Q16 ? (unsigned int)reinterpret_cast<const unsigned short *>(QUEUE)[64 + lane] : QUEUE[64 + lane];
        if (act) {
            const unsigned int e = Q16 ? (unsigned int)reinterpret_cast<const unsigned short *>(QUEUE)[lane] : QUEUE[lane];
            kk = kbase + (int)(e >> kshift);
            bb = (int)(e & kmask);
            const int ss = LOCAL ? kk : sample_of(a, kk);
            const BoxRec<T> bx = boxes[bb];
            const T *rs = REC + ss * NREC, *rk = REC + kk * NREC;
            const T qz = rs[kP2];
            T qx = rs[kP0], qy = rs[kP1], qc = rk[kCC], qs = rk[kSS], bcx = bx.cx, bcy = bx.cy;
            if (ROT == 2 && box_is_tilted(bx)) {   // general rotation: the (large, rare) exact OBB routine
                key = tilt_key(qx, qy, qz, qc, qs, a.ax, a.ay, a.az, bx, a.tilt);
            } else {
                if (ROT) {   // evaluate the pair in the box's own frame
                    to_box_frame(bx, qx, qy, qc, qs);
                    bcx = T(0); bcy = T(0);
                }
                Pose<T> ps;
                pose_setup(ps, qx, qy, qz, qc, qs, a.ax, a.ay);
                key = pair_key(ps, a.ax, a.ay, a.az, bcx, bcy, bx.cz, bx.hx, bx.hy, bx.hz);
            }
#ifdef MP_STATS
            if (ROT == 0) {   // how many evaluated pairs needed the rectangle distance (xy-separated), how many were overlaps
                Pose<T> pst;
                pose_setup(pst, qx, qy, qz, qc, qs, a.ax, a.ay);
                PairOut<T> po;
                pair_axes(pst, a.ax, a.ay, a.az, bcx, bcy, bx.cz, bx.hx, bx.hy, bx.hz, po);
                if (po.gsat > T(0)) atomicAdd(&g_stats[5], 1ull);
                if (!(po.gsat > T(0)) && !(po.gz > T(0))) atomicAdd(&g_stats[6], 1ull);
            }
#endif
            if (ARGS) oldk = MINKEY[kk * KS];
        }
        if (rest > 0) {   // the queue reads of this pass were issued above; LDS operations of a wave complete in issue order
            wave_lds_sync();
            if (lane < rest) {
                if (Q16) reinterpret_cast<unsigned short *>(QUEUE)[lane] = (unsigned short)moved;
                else QUEUE[lane] = moved;
            }
        }
        if (ARGS) {
            wave_lds_sync();
            if (act) atomicMin(&MINKEY[kk * KS], key);
            wave_lds_sync();
            bool win = false;
            if (act) {
                const unsigned long long newk = MINKEY[kk * KS];
                win = key == newk;
                if (win && newk < oldk) ARG[kk * AS] = 0x7fffffff;  // strictly better: restart the argmin
            }
            wave_lds_sync();
            if (win) atomicMin(&ARG[kk * AS], bb);  // ties: lowest box index, like a first-minimum scan
            wave_lds_sync();
        } else {
            if (act) atomicMin(&MINKEY[kk * KS], key);
            wave_lds_sync();
        }
        if (valid) {
            const unsigned long long kcur = MINKEY[own * KS];
            U32 = __builtin_fminf(U32, (float)key_to_sep2<T>(kcur) * (1.0f + 4e-6f));   // never above the seed
            if (!(kcur & kSepBit)) {
                // penetrating: shrink by the depth, rounded down (4e-6 relative covers the conversion to fp32)
                const float dsh = (float)Key<T>::from_bits(Key<T>::kInfBits - kcur) * (1.0f - 4e-6f);
                mlox = (px32 - exs) + dsh; mloy = (py32 - eys) + dsh; mloz = (pz32 - ezs) + dsh;
                mhix = (px32 + exs) - dsh; mhiy = (py32 + eys) - dsh; mhiz = (pz32 + ezs) - dsh;
            }
        }
    };

    auto push = [&](bool need, int idx) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(need);
        if (m) {
            if (need) {
                const int pos = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, (unsigned int)qcount));
                if (Q16) reinterpret_cast<unsigned short *>(QUEUE)[pos] = (unsigned short)(lane_hi | (unsigned int)idx);
                else QUEUE[pos] = lane_hi | (unsigned int)idx;
            }
            qcount += __builtin_popcountll(m);
            if (qcount >= 64) {
                drain(64, qcount - 64);
                qcount -= 64;
            }
        }
    };

    // Wide box first (a table, a floor; chosen on the host, axis-aligned scenes only, record 0 of the table).  Where
    // the mover's centre lies over its footprint the two footprints overlap for certain (every 2-D SAT gap is at most
    // minus the mover's smaller half extent), so with a positive z gap the pair value is gz^2, computed exactly as
    // pair_key would, and the lane settles the pair on its own sample slot; the other lanes queue it as usual
    // through the queue.
    int wide_bi = -1;
    if (!ROT && n_wide > 0) {
        const BoxRec32 bx = load_box32_uniform(boxes32, 0);
        const int bi = __float_as_int(bx.pad0);
        wide_bi = bi;
        const BoxRec<T> eb = load_box_uniform(boxes, bi);
        bool need = aabb_lb2(mlox, mloy, mloz, mhix, mhiy, mhiz, bx) <= U32;
        if (need && a.ax > T(1e-3) && a.ay > T(1e-3)) {
            const int sw = LOCAL ? lane : sample_of(a, k0 + lane);
            const T *rw = REC + sw * NREC;
            const T gz = mp_abs(rw[kP2] - eb.cz) - (a.az + eb.hz);
            if (mp_abs(rw[kP0] - eb.cx) <= eb.hx && mp_abs(rw[kP1] - eb.cy) <= eb.hy && gz > T(0)) {
                MINKEY[own * KS] = kSepBit | Key<T>::bits(mp_fma(gz, gz, T(0)));   // the slot is still at its initial value
                if (ARGS) ARG[own * AS] = bi;
                U32 = __builtin_fminf(U32, (float)key_to_sep2<T>(MINKEY[own * KS]) * (1.0f + 4e-6f));
                need = false;
            }
        }
        push(need, bi);
    }


    // Per-sample seed of the bound: a box that is likely to be the sample's nearest, evaluated exactly, first, for every lane at
    // once (one full exact pass).  It bounds the lane's clearance from above before the sweep starts, so the sweep queues only
    // boxes that can still beat a near-final value; without it every box the sweep meets before a lane's first exact pass is
    // queued against an infinite bound (measured with a perfect seed: 249 -> 73 exact pairs per candidate at config 3,
    // 137 -> 49 per chunk at B=13 S=101, 779 -> 131 at config 4).  Sources: the box that decides this collision sample on the
    // MEAN trajectory of the population the candidate was drawn from (mp_api.hip, cem_solve), else the box nearest to the
    // centre of the cell the sample lies in, looked up in the uniform grid the host builds per box range (mp_api.hip,
    // upload_problem).  The result does not depend on the seed: the sweep still queues every box whose bound is not above
    // the lane's value.
    int sb = -1;
    if (seeded) {
        if (seed_row) {
            if (valid) sb = seed_row[k0 + lane];
        } else {
            const int gnx = ldu(&tk.gnx), gny = ldu(&tk.gny), gnz = ldu(&tk.gnz);
            const float fx = (px32 - ldu(&tk.gox)) * ldu(&tk.gix), fy = (py32 - ldu(&tk.goy)) * ldu(&tk.giy),
                        fz = (pz32 - ldu(&tk.goz)) * ldu(&tk.giz);
            // clamped in floating point first (a far-away or NaN sample must not overflow the conversion)
            const int ix = (int)__builtin_fminf(__builtin_fmaxf(fx, 0.f), (float)(gnx - 1));
            const int iy = (int)__builtin_fminf(__builtin_fmaxf(fy, 0.f), (float)(gny - 1));
            const int iz = (int)__builtin_fminf(__builtin_fmaxf(fz, 0.f), (float)(gnz - 1));
            if (valid) sb = a.grid[grid_begin + (iz * gny + iy) * gnx + ix];
        }
        push(sb >= 0 && sb != wide_bi, sb);   // (the wide box has been settled or queued above for every lane)
        if (qcount > 0) {
            drain(qcount, 0);
            qcount = 0;
        }
    }

    // Bound pass over the task's boxes in sorted order.  With several groups a whole group
    // is skipped when no lane of the wave can be closer to its AABB than its current minimum.
    // (two copies of the sweep: only a seeded launch pays for the comparison with the seed box)
    auto sweep = [&](auto with_seed) {
    constexpr bool WS = decltype(with_seed)::value;
    auto visit = [&](int g) {
        const BoxRec32 grp = load_box32_uniform(groups, 2 * g);
        MP_STAT_ADD(7, 1);
        const int gfirst = __float_as_int(grp.pad0), gcnt = __float_as_int(grp.pad1);
        if (ngroups > 1) {
            const float lbg = aabb_lb2(mlox, mloy, mloz, mhix, mhiy, mhiz, grp);
            if (!__builtin_amdgcn_ballot_w64(lbg <= U32)) return;
        }
        MP_STAT_ADD(1, 1); MP_STAT_ADD(2, gcnt);
        // two boxes per trip (packed fp32): group j of 16 records is pairs 8j ... of the task's pair table
        const int pfirst = (gfirst - n_wide) >> 1, npairs = (gcnt + 1) >> 1;
        // two records in flight, used alternately (unrolled by two: the record being tested and the one being fetched swap
        // roles instead of being copied from one set of scalar registers to the other every trip: config-4 shape -3.7 %)
        auto test2 = [&](const BoxPair32 &bp) {
            const v2f lb2 = aabb_lb2_pair(mlox, mloy, mloz, mhix, mhiy, mhiz, bp);
            push(lb2.x <= U32 && (!WS || bp.idx0 != sb), bp.idx0);   // the seed box has been evaluated
            push(bp.idx1 >= 0 && lb2.y <= U32 && (!WS || bp.idx1 != sb), bp.idx1);   // against the bound as the first push may have refreshed it
        };
        BoxPair32 ra = load_pair_uniform(pairs, pfirst);
        for (int i = 0; i < npairs; i += 2) {
            const BoxPair32 rb = load_pair_uniform(pairs, pfirst + (i + 1 < npairs ? i + 1 : i));
            test2(ra);
            ra = load_pair_uniform(pairs, pfirst + (i + 2 < npairs ? i + 2 : i));
            if (i + 1 < npairs) test2(rb);
        }
    };
    if (SEED && ngroups > 1) {
        // Group prefilter (the split kernel; scenes large enough to have groups): 64 groups per trip, one per LANE, against
        // the one region every lane's search is confined to - the union over the wave's samples of the mover's AABB
        // inflated by r = sqrt(U32) - and only the groups that touch it get the per-sample test above.  Conservative: a
        // group some lane needs has lb2 <= U32 there, so along every axis its gap to that lane's AABB is at most
        // sqrt(U32) (each squared gap is one of the non-negative terms of lb2, and fp32 addition of a non-negative term
        // never lowers a sum), i.e. the group's AABB touches [mlo - r, mhi + r]; r carries 1e-5 relative and the
        // slack on top, far more than the roundings of the additions below (2.4e-7 of the coordinates).  The bounds
        // only tighten while the sweep runs, so a region taken here stays valid (if no longer minimal) to its end.
        const float rr = valid ? __builtin_fmaf(__builtin_sqrtf(__builtin_fmaxf(U32, 0.f)), 1.0f + 1e-5f, slack) : 0.f;
        const float inf = __builtin_huge_valf();
        const float ulox = wave_min(valid ? mlox - rr : inf), uloy = wave_min(valid ? mloy - rr : inf),
                    uloz = wave_min(valid ? mloz - rr : inf);
        const float uhix = -wave_min(valid ? -(mhix + rr) : inf), uhiy = -wave_min(valid ? -(mhiy + rr) : inf),
                    uhiz = -wave_min(valid ? -(mhiz + rr) : inf);
        for (int g0 = 0; g0 < ngroups; g0 += 64) {
            const int gi = g0 + lane;
            bool hit = false;
            if (gi < ngroups) {
                const BoxRec32 &gr = groups[2 * gi];
                hit = gr.lox <= uhix && gr.hix >= ulox && gr.loy <= uhiy && gr.hiy >= uloy && gr.loz <= uhiz && gr.hiz >= uloz;
            }
            unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
            while (m) {
                const int g = g0 + __builtin_ctzll(m);
                m &= m - 1;
                visit(g);
            }
        }
    } else {
        for (int g = 0; g < ngroups; ++g) visit(g);
    }
    };
    if (seeded) sweep(std::true_type{});
    else sweep(std::false_type{});
    // What is left goes into an exact pass of its own, unless another chunk of the trajectory follows: then the entries
    // stay queued and share a pass with that chunk's first entries (S = 101: 64 + 37 samples, four partly filled passes
    // become three).
    if (keep_tail) qcarry = qcount;
    else {
        if (qcount > 0) drain(qcount, 0);
        qcarry = 0;
    }
    wave_lds_sync();
}

// ROT: 0 axis-aligned scene; 1 the scene holds yaw-rotated static boxes; 2 it also holds tilted ones (separate
// instantiations keep the axis-aligned code as it was)
// PRE: split mode, the clearance profile comes from clearance_kernel (box model only)
// Waves per SIMD the register budget is sized for: fp64 gradient kernels 4 (122 VGPRs), fp64 cost-only kernels 5 (95; round 1 needed 168
// for the gradient kernels: 3 waves), fp32 kernels 7 (72 VGPRs).  No scratch memory except one 8-byte accumulator per
// lane in the fused fp64 cost-only box instances (profiles/r02_kernel_resources.txt).
template <typename T, bool GRAD, int ROT = 0, int WV = 0, int NW = 1> constexpr int eval_min_waves()
{
    if (WV > 0) return WV;
    if (NW > 1) return sizeof(T) == 4 ? MP_F32_WAVES : MP_WIDE_WAVES;
    return ROT == 2 ? 2   // the tilt routine is a real call: its registers count too
                    : (sizeof(T) == 4 ? MP_F32_WAVES : (GRAD ? MP_GRAD64_WAVES : MP_COST64_WAVES));
}

// WV = 5: the five-waves-per-SIMD instance of the fp64 gradient kernel (fused, axis-aligned box scenes, one 64-sample
// chunk, < 1024 boxes per task, <= 8 waypoints): 96 VGPRs (~70 B of scratch per lane) and a slice of 15 * 8 * S + 256 bytes,
// so that five four-wave blocks fit the CU's 128 LDS granules of 1,280 bytes at S = 64.  What it gives up to get there:
//   * the waypoint table W shares its 256 bytes with the work queue (16-bit entries) and, after the clearance phase, with
//     the deciding boxes as 16-bit values; the closed-form terms read the free waypoints from x itself;
//   * during the exact passes the 32-bit deciding-box words (they need 32-bit LDS atomics) live in the low half of the
//     records' kD slot, which is free until the clearance is written there at the end of the chunk.
// NW = 3 | 4: the wide instance (split-mode gradient launches without term breakdown on trajectories of more than 176
// samples): one candidate per block of NW waves, launched with 64 * NW threads per block and the standard slice + NW * 64
// values (RED) of dynamic LDS; everything else — phases, layout, arithmetic per sample — is the NW = 1 code.
template <typename T, int MODEL, bool GRAD, int ROT, bool PRE, int WV = 0, int NW = 1>
__global__ void __launch_bounds__(256, (eval_min_waves<T, GRAD, ROT, WV, NW>())) eval_kernel(const EvalArgs<T> a_by_value)
{
    // The arguments are read through the kernel-argument segment pointer (the explicit arguments start at offset 0),
    // re-derived behind an opaque barrier at the start of every phase of the candidate loop (MP_ARGS_REFRESH).  Read as
    // plain by-value fields, all ~120 dwords of EvalArgs are loaded once and held in scalar registers across the whole
    // loop; there are 102 of those, so the compiler parked the excess in VGPR lanes (v_writelane once, ~450
    // v_readlane per candidate: one in seven vector instructions of the kernel).  A phase now holds only what it
    // uses; the re-reads are scalar loads that hit the scalar cache.
    typedef const EvalArgs<T> __attribute__((address_space(4))) KArgs;
    KArgs *kargs = (KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    (void)a_by_value;
#define a (*kargs)
    // `lane` goes through the same barrier: what the compiler derives from it (record addresses, masks) would otherwise
    // be computed once and held in vector registers across the whole loop (163 VGPRs; recomputed per phase: 123)
#define MP_ARGS_REFRESH() asm volatile("" : "+s"(kargs), "+v"(lane))
    // the scalar arguments a phase may use, read together at its start (one batch of scalar loads; what a phase does not
    // use is dropped by the compiler); `t` = the candidate's task
#define MP_LOCALS                                                                                                          \
    const int S = a.S, C = a.C, Cy = a.Cy, nf = a.nf, ff = a.ff, nyf = a.nyf, D = 3 * nf + nyf;                            \
    const unsigned mask = a.mask;                                                                                          \
    const int keep_n = a.keep_n, use_pref = a.use_pref, tail_start = a.tail_start, use_goal_target = a.use_goal_target,    \
              goal_tail_n = a.goal_tail_n, last_cond = a.last_cond;                                                        \
    const T du = a.du, inv_du = a.inv_du, inv_h2 = a.inv_h2, req = a.req, pref = a.pref, appr_tgt = a.appr_tgt,            \
            goal_tgt = a.goal_tgt;                                                                                         \
    T w[kNTerms];                                                                                                          \
    _Pragma("unroll") for (int wi_ = 0; wi_ < kNTerms; ++wi_) w[wi_] = a.w[wi_];                                           \
    T w2[kNTerms];                                                                                                         \
    _Pragma("unroll") for (int wi_ = 0; wi_ < kNTerms; ++wi_) w2[wi_] = a.w2[wi_];                                         \
    T wf[kNTerms];                                                                                                         \
    _Pragma("unroll") for (int wi_ = 0; wi_ < kNTerms; ++wi_) wf[wi_] = a.wf[wi_];                                         \
    const T *M = a.M, *My = a.My, *tsched = a.tsched;                                                                      \
    const uint8_t *amask = a.amask;                                                                                        \
    const DevTask<T> &tk = a.tasks[t];                                                                                     \
    const BoxRec<T> *boxes = a.boxes + ldu(&tk.box_begin);                                       \
    (void)S; (void)C; (void)Cy; (void)nf; (void)ff; (void)nyf; (void)D; (void)mask; (void)keep_n; (void)use_pref;          \
    (void)tail_start; (void)use_goal_target; (void)goal_tail_n; (void)last_cond; (void)du; (void)inv_du; (void)inv_h2;     \
    (void)req; (void)pref; (void)appr_tgt; (void)goal_tgt; (void)w; (void)w2; (void)wf; (void)M; (void)My; (void)tsched; (void)amask;          \
    (void)tk; (void)boxes
    extern __shared__ __align__(16) unsigned char smem_raw[];
    int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const int S_layout = a.S;   // inside the candidate loop every argument is re-read per phase (MP_LOCALS)
    // WV > 0: a compact instance (no term breakdown, W shares the queue's 256 bytes, 16-bit queue entries); FIVE: its
    // gradient form (deciding boxes in the records / as 16-bit values, one chunk); the cost-only form keeps no deciding boxes
    constexpr bool COMPACT = WV > 0;
    constexpr bool FIVE = COMPACT && GRAD;
    static_assert(!COMPACT || (sizeof(T) == 8 && MODEL == 1 && !PRE), "the compact instances are fused fp64 box-model kernels");
    // NW > 1: the wide instance (split mode, long trajectories): ONE candidate per block, sample s on lane s mod 64 of wave
    // (s / 64) mod NW, so that a 256-sample trajectory keeps four waves busy per 32 KB slice instead of one; the phases
    // are separated by block barriers, sums cross the waves through RED (NW x 64 values behind the slice)
    constexpr bool WIDE = NW > 1;
    constexpr bool FOLD = COMPACT || WIDE;   // no term breakdown: the weighted partial sums are folded into `part` phase by phase
    static_assert(!WIDE || (PRE && MODEL == 1 && WV == 0 && GRAD), "the wide instance is a split-mode box-model gradient kernel");
    const int wv = WIDE ? wib : 0;       // which 64-sample column of every trip this wave takes
    const int slice = WIDE ? 0 : wib;
#define MP_CAND_SYNC() do { if constexpr (NW > 1) __syncthreads(); else wave_lds_sync(); } while (0)
    constexpr int NREC = rec_fields(GRAD);
    T *base = reinterpret_cast<T *>(smem_raw + (size_t)slice * a.lds_per_wave);
    T *RED = reinterpret_cast<T *>(smem_raw + a.lds_per_wave);
    (void)RED;
    T *REC = (COMPACT || WIDE) ? base : base + a.w_elems;
#define RF(s_, f_) REC[(s_) * NREC + (f_)]
    // ordered clearance keys: field kGD of the records (fp64 gradient launches: stride NREC keys) or an array behind them
    constexpr bool KEYREC = key_in_record(GRAD, (int)sizeof(T));
    constexpr int KS = KEYREC ? NREC : 1;
    unsigned char *behind = smem_raw + (size_t)slice * a.lds_per_wave + (((((COMPACT || WIDE) ? 0 : a.w_elems) + NREC * S_layout) * sizeof(T) + 7) & ~(size_t)7);
    unsigned long long *MINKEY = KEYREC ? reinterpret_cast<unsigned long long *>(REC + (GRAD ? (int)kGD : 0))
                                        : reinterpret_cast<unsigned long long *>(behind);
    // deciding boxes: an int array behind the records; five-wave instance: the low half of the records' kD slot during the
    // exact passes (stride 2 * NREC ints), then 16-bit values in the queue's memory (ARG16)
    constexpr int AS = FIVE ? 2 * NREC : 1;
    int *ARG = FIVE ? reinterpret_cast<int *>(REC + (int)kD) : reinterpret_cast<int *>(behind + ((KEYREC || WIDE) ? 0 : 8 * (size_t)S_layout));   // (wide: split mode, no keys)
    // compact cost-only form: MINKEY[S] | QW behind the records
    unsigned int *QUEUE = FIVE ? reinterpret_cast<unsigned int *>(behind)
                               : reinterpret_cast<unsigned int *>(ARG + (GRAD ? S_layout : 0));
    short *ARG16 = reinterpret_cast<short *>(QUEUE);
    // wide instance: W is read by the spline pass only (the closed-form terms read x itself) and shares RED's memory, which
    // is written after the barriers that follow that pass
    T *W = COMPACT ? reinterpret_cast<T *>(QUEUE) : (WIDE ? RED : base);
    const int RS = WIDE ? a.wide_rs : 64;   // distance, in values, between the waves' shares of the M^T g outputs in RED
    (void)ARG16;
    MP_PHASE_DECL;
    for (long long cand = WIDE ? (long long)blockIdx.x : (long long)blockIdx.x * wpb + wib; cand < a.n;
         cand += WIDE ? (long long)gridDim.x : (long long)gridDim.x * wpb) {
        MP_ARGS_REFRESH();
        long long tl, grp = 0;
        if (a.group_size > 0) {
            grp = cand / a.group_size;
            if (a.group_done && a.group_done[grp]) continue;
            tl = a.group_task ? (long long)a.group_task[grp] : 0;
        } else {
            tl = a.task_of ? (long long)a.task_of[cand] : (a.cpt > 0 ? cand / a.cpt : 0);
        }
        if (tl < 0 || tl >= (long long)a.n_tasks) {
            // a task index outside the problem (device-resident task_of is not checked on the host): no table is
            // touched, the candidate gets NaN results and no flag
            const T nan = Lim<T>::inf() - Lim<T>::inf();
            if (lane == 0) {
                a.cost[cand] = nan;
                if (a.minc) a.minc[cand] = nan;
                if (a.flags) a.flags[cand] = 0;
            }
            if (a.terms && lane < kNTerms) a.terms[cand * kNTerms + lane] = nan;
            const int D = 3 * a.nf + a.nyf;
            if (GRAD) for (int o = lane; o < D; o += 64) a.grad[cand * D + o] = nan;
            if (a.dist) for (int k = lane; k < a.Sp; k += 64) a.dist[cand * a.Sp + k] = nan;
            continue;
        }
        const int t = __builtin_amdgcn_readfirstlane((int)tl);
        // state that crosses the phases below
        T t_len = T(0), t_ys = T(0), t_mono = T(0), t_sched = T(0), t_curv = T(0);
        T t_safe = T(0), t_pref = T(0), t_reb = T(0), t_gc = T(0), t_gt = T(0), t_ac = T(0), t_acol = T(0);
        T t_via = T(0), t_ydev = T(0), t_gn = T(0), gnx = T(0), gny = T(0), gnz = T(0);
        T mc = Lim<T>::inf();
        T part = T(0);   // five-wave instance: the weighted per-lane partial sum, folded phase by phase (no breakdown there)
        int Sp = 0, gn_first = 0;

        MP_PHASE_INIT;
        {
        MP_LOCALS;
        const T *x = a.X + cand * D;
        if (C == 4 && Cy == 4 && ff == 1 && nf == 2 && nyf == 2) {
            // the reference's default layout: waypoints in registers / scalar operands, no table (spline_sample4)
            T xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] = x[i];
            const TaskEnds<T> te = load_task_ends<T>(tk);
            for (int s = lane + 64 * wv; s < S; s += 64 * NW) {
                T px, py, pz, yw;
                spline_sample4(M + (size_t)s * 4, My + (size_t)s * 4, te, xv, px, py, pz, yw);
                RF(s, kP0) = px; RF(s, kP1) = py; RF(s, kP2) = pz; RF(s, kYW) = yw;
            }
        } else {
            // ---- waypoint table W[j][0..2] = position rows, W[j][3] = yaw controls ----
            const int rows = C > Cy ? C : Cy;
            for (int i = lane + 64 * wv; i < rows * 4; i += 64 * NW) {
                const int j = i >> 2, col = i & 3;
                T v = T(0);
                if (col < 3) {
                    if (j < C) v = (j >= ff && j < ff + nf) ? x[3 * (j - ff) + col] : tk.ctrl[j][col];
                } else if (j < Cy) {
                    v = (j >= 1 && j <= nyf) ? x[3 * nf + (j - 1)] : tk.yaw_ctrl[j];
                }
                W[i] = v;
            }
            MP_CAND_SYNC();
            // ---- spline samples Q = M W (fixed fma chain over the waypoints) ----
            for (int s = lane + 64 * wv; s < S; s += 64 * NW) {
                T px, py, pz, yw;
                spline_sample(M + (size_t)s * C, My + (size_t)s * Cy, C, Cy, W, px, py, pz, yw);
                RF(s, kP0) = px; RF(s, kP1) = py; RF(s, kP2) = pz; RF(s, kYW) = yw;
            }
        }
        MP_CAND_SYNC();

        }
        MP_PHASE(0);  // waypoints + spline
        MP_ARGS_REFRESH();
        {
        MP_LOCALS;
        // ---- per-sample geometric terms, pass 1 ----
        const T y0 = ldu(&tk.yaw_ctrl[0]), y1 = ldu(&tk.yaw_ctrl[Cy - 1]);
        const int yaw_inc = ldu(&tk.yaw_inc);
        for (int s = lane + 64 * wv; s < S; s += 64 * NW) {
            const T px = RF(s, kP0), py = RF(s, kP1), pz = RF(s, kP2), yw = RF(s, kYW);
            if (s + 1 < S) {
                T ex = RF(s + 1, kP0) - px, ey = RF(s + 1, kP1) - py, ez = RF(s + 1, kP2) - pz;
                T n, inv_n;
                sqrt_rsqrt(mp_fma(ez, ez, mp_fma(ey, ey, ex * ex)), n, inv_n);
                t_len += n;
                if (GRAD) RF(s, kSEGW) = n > T(0) ? w[0] * inv_n : T(0);  // w_len / |segment s|, reused by the reverse sweep
                T dy = RF(s + 1, kYW) - yw;
                T e = npmax0(yaw_inc ? -dy : dy);
                t_mono = mp_fma(e, e, t_mono);
            }
            if (s + 2 < S) {
                T e = mp_fma(T(-2), RF(s + 1, kYW), yw) + RF(s + 2, kYW);   // (2 y is exact: the same value as y - 2 y' + y'')
                t_ys = mp_fma(e, e, t_ys);
            }
            {
                T e = yw - mp_fma(y1 - y0, tsched[s], y0);
                t_sched = mp_fma(e, e, t_sched);
            }
            if (S >= 3) {  // np.gradient(P, du): one-sided at the ends, central inside
                const int lo = s == 0 ? 0 : s - 1, hi = s == S - 1 ? s : s + 1;
                const T sc = (s == 0 || s == S - 1) ? inv_du : inv_h2;
                RF(s, kD1x) = (RF(hi, kP0) - RF(lo, kP0)) * sc; RF(s, kD1y) = (RF(hi, kP1) - RF(lo, kP1)) * sc; RF(s, kD1z) = (RF(hi, kP2) - RF(lo, kP2)) * sc;
            }
        }
        if (FOLD) {   // same order of additions as the cost-only reduction below
            part = mp_fma(wf[0], t_len, part);
            part = mp_fma(wf[2], t_ys, part);
            part = mp_fma(wf[12], t_mono, part);
            part = mp_fma(wf[13], t_sched, part);
        }
        MP_CAND_SYNC();
        }
        MP_PHASE(1);  // pass 1
        MP_ARGS_REFRESH();
        {
        MP_LOCALS;
        // ---- curvature, pass 2 (second np.gradient) ----
        if ((mask & (1u << 1)) && S >= 3) {
            const T eps = T(1e-9);
            const T kc1 = w[1] * du, kc2 = kc1 * T(2), kc5 = kc1 * T(-5);
            (void)kc2; (void)kc5;
            for (int s = lane + 64 * wv; s < S; s += 64 * NW) {
                const T v1x = RF(s, kD1x), v1y = RF(s, kD1y), v1z = RF(s, kD1z);
                const int lo = s == 0 ? 0 : s - 1, hi = s == S - 1 ? s : s + 1;
                const T sc = (s == 0 || s == S - 1) ? inv_du : inv_h2;
                const T v2x = (RF(hi, kD1x) - RF(lo, kD1x)) * sc, v2y = (RF(hi, kD1y) - RF(lo, kD1y)) * sc, v2z = (RF(hi, kD1z) - RF(lo, kD1z)) * sc;
                T speed, inv_speed;
                sqrt_rsqrt(mp_fma(v1z, v1z, mp_fma(v1y, v1y, v1x * v1x)), speed, inv_speed);
                // c = d1 x d2, every component one product and one fma
                T cx = mp_fma(v1y, v2z, -(v1z * v2y)), cy = mp_fma(v1z, v2x, -(v1x * v2z)), cz = mp_fma(v1x, v2y, -(v1y * v2x));
                T cn2 = mp_fma(cz, cz, mp_fma(cy, cy, cx * cx));
                // kappa^2 speed = |c|^2 speed / max(speed, eps)^6 (|c| itself is never needed)
                const T inv_sp = speed > eps ? inv_speed : T(1e9);
                const T inv_sp3 = inv_sp * inv_sp * inv_sp;
                const T inv_sp6 = inv_sp3 * inv_sp3;
                const T gs0 = cn2 * inv_sp6;          // |c|^2 / sp^6
                t_curv = mp_fma(gs0, speed, t_curv);
                if (GRAD) {
                    // f = |c|^2 speed / sp^6 with c = d1 x d2; adjoints of d1 and d2 (kc2 = 2 w_curv du)
                    const T f2 = (kc2 * speed) * inv_sp6;
                    const T gcx = f2 * cx, gcy = f2 * cy, gcz = f2 * cz;
                    T q = T(0);
                    if (speed > eps) q = (kc5 * gs0) * inv_sp;          // explicit + through sp (= speed): (1 - 6) |c|^2 / sp^6, kc5 = -5 w_curv du
                    else if (speed > T(0)) q = (kc1 * gs0) * inv_speed;
                    RF(s, kA1x) = mp_fma(v2y, gcz, mp_fma(-v2z, gcy, q * v1x));
                    RF(s, kA1y) = mp_fma(v2z, gcx, mp_fma(-v2x, gcz, q * v1y));
                    RF(s, kA1z) = mp_fma(v2x, gcy, mp_fma(-v2y, gcx, q * v1z));
                    RF(s, kA2x) = mp_fma(gcy, v1z, -(gcz * v1y)); RF(s, kA2y) = mp_fma(gcz, v1x, -(gcx * v1z)); RF(s, kA2z) = mp_fma(gcx, v1y, -(gcy * v1x));
                }
            }
        }
        MP_CAND_SYNC();   // the clearance phase reuses the D1 fields (CC SS D)
        if (FOLD) part = mp_fma(wf[1], t_curv, part);

        }
        MP_PHASE(2);  // curvature
        MP_ARGS_REFRESH();
        {
        MP_LOCALS;
        // ---- clearance profile d_safe ----
        Sp = a.Sp;
        if (MODEL == 1 && PRE) {
            // split mode: the clearance profile was computed by clearance_kernel (one wave per 64-sample chunk)
            for (int k = lane + 64 * wv; k < Sp; k += 64 * NW) {
                const T d = a.pre_d[cand * a.Sp + k];
                RF(k, kD) = d;
                if (GRAD) ARG[k] = a.pre_arg[cand * a.Sp + k];
                if (GRAD) {
                    T c, sn;
                    yaw_rot(RF(sample_of(a, k), kYW), a.trig, c, sn);
                    RF(k, kCC) = c; RF(k, kSS) = sn;
                }
                if (a.dist) a.dist[cand * a.Sp + k] = d;
            }
        } else if (MODEL == 1) {
            const BoxRec32 *boxes32 = a.boxes32 + ldu(&tk.sbox_begin);
            const BoxRec32 *groups = a.groups + 2 * ldu(&tk.group_begin);
            const int ngroups = ldu(&tk.group_count);
            const int n_wide = ldu(&tk.n_wide);
            // entries of different chunks may share an exact pass when a queue entry has room for the trajectory's sample index
            const bool defer_ok = Sp > 64 && (COMPACT ? a.defer == 1 : a.defer != 0);
            int qcarry = 0;
            for (int k0 = 0; k0 < Sp; k0 += 64)
                clearance_chunk<T, ROT, false, false, NREC, KS, AS, COMPACT, GRAD, (ROT < 2)>(a, tk, boxes, boxes32, a.pairs + ldu(&tk.spair_begin), groups,
                                                                          ngroups, n_wide, lane, k0, Sp, REC, MINKEY, ARG, QUEUE,
                                                                          a.seed ? a.seed + (size_t)grp * Sp : nullptr, qcarry,
                                                                          defer_ok, defer_ok && k0 + 64 < Sp);
            for (int k0 = 0; k0 < Sp; k0 += 64) {
                const int k = k0 + lane;
                if (k < Sp) {
                    const T d = key_to_distance<T>(MINKEY[k * KS]);
                    if (FIVE) {   // out of the kD slot (same lane) before d goes in; the queue is dead (one chunk)
                        const int ag = *reinterpret_cast<volatile int *>(&ARG[k * AS]);
                        ARG16[k] = (short)ag;
                        __asm__ volatile("" ::: "memory");   // the int read and the double store alias
                    }
                    RF(k, kD) = d;
                    if (a.dist) a.dist[cand * a.Sp + k] = d;
                }
            }
        } else {
            for (int k = lane + 64 * wv; k < Sp; k += 64 * NW) {
                const int s = sample_of(a, k);
                int arg = -1;
                const T d = point_distance(RF(s, kP0), RF(s, kP1), RF(s, kP2), boxes, ldu(&tk.box_count), a.radius, ROT != 0, ROT == 2 ? a.tilt : nullptr, arg);
                RF(k, kD) = d;
                if (GRAD) ARG[k] = arg;
                if (a.dist) a.dist[cand * a.Sp + k] = d;
            }
        }
        MP_CAND_SYNC();
        if (last_cond && Sp >= 2) {  // P[::k] + conditional last sample (spline_opt.py:802-805)
            const int sa = sample_of(a, Sp - 2), sb = S - 1;
            bool close = mp_abs(RF(sa, kP0) - RF(sb, kP0)) <= T(1e-8) + T(1e-5) * mp_abs(RF(sb, kP0));
            close = close && (mp_abs(RF(sa, kP1) - RF(sb, kP1)) <= T(1e-8) + T(1e-5) * mp_abs(RF(sb, kP1)));
            close = close && (mp_abs(RF(sa, kP2) - RF(sb, kP2)) <= T(1e-8) + T(1e-5) * mp_abs(RF(sb, kP2)));
            if (close) Sp -= 1;
        }

        }
        MP_PHASE(3);  // clearance profile
        MP_ARGS_REFRESH();
        {
        MP_LOCALS;
        // ---- clearance terms (and d cost / d d_k) ----
        for (int k = lane + 64 * wv; k < Sp; k += 64 * NW) {
            const T d = RF(k, kD);
            mc = d < mc ? d : mc;
            T gd = T(0);
            // (w2 = 2 w: d(e^2) = 2 e de; every update of gd is one fma)
            if (mask & (1u << 3)) {
                T e = npmax0(req - d);
                t_safe = mp_fma(e, e, t_safe);
                if (GRAD) gd = mp_fma(-w2[3], e, gd);
            }
            if ((mask & (1u << 4)) && use_pref && k < keep_n) {
                T e = npmax0(pref - d);
                t_pref = mp_fma(e, e, t_pref);
                if (GRAD) gd = mp_fma(-w2[4], e, gd);
            }
            if (mask & (1u << 5)) {
                if (k >= tail_start && k + 1 < Sp) {
                    T e = npmax0(RF(k + 1, kD) - d);
                    t_reb = mp_fma(e, e, t_reb);
                    if (GRAD) gd = mp_fma(-w2[5], e, gd);
                }
                if (GRAD && k - 1 >= tail_start && k - 1 >= 0) {
                    T e = npmax0(d - RF(k - 1, kD));
                    gd = mp_fma(w2[5], e, gd);
                }
            }
            if (k == Sp - 1) {
                if (mask & (1u << 6)) {
                    T e = d - pref;
                    e = e > T(0) ? e : T(0);
                    t_gc = e * e;
                    if (GRAD) gd = mp_fma(w2[6], e, gd);
                }
                if ((mask & (1u << 7)) && use_goal_target) {
                    T e = d - goal_tgt;
                    t_gt = e * e;
                    if (GRAD) gd = mp_fma(w2[7], e, gd);
                }
            }
            if (amask[k]) {
                if (mask & (1u << 8)) {
                    T e = npmax0(appr_tgt - d);
                    t_ac = mp_fma(e, e, t_ac);
                    if (GRAD) gd = mp_fma(-w2[8], e, gd);
                }
                if (mask & (1u << 9)) {
                    T e = npmax0(-d);
                    t_acol = mp_fma(e, e, t_acol);
                    if (GRAD) gd = mp_fma(-w2[9], e, gd);
                }
            }
            if (GRAD) RF(k, kGD) = gd;
        }
        if (FOLD) {
            part = mp_fma(wf[3], t_safe, part);
            part = mp_fma(wf[4], t_pref, part);
            part = mp_fma(wf[5], t_reb, part);
            part = mp_fma(wf[6], t_gc, part);
            part = mp_fma(wf[7], t_gt, part);
            part = mp_fma(wf[8], t_ac, part);
            part = mp_fma(wf[9], t_acol, part);
        }

        }
        MP_PHASE(4);  // clearance terms
        MP_ARGS_REFRESH();
        {
        MP_LOCALS;
        // ---- deviation terms: lane o < D holds (x_o - its initial value)^2 of optimisation variable o (the fixed yaw controls
        // of the reference's full sum contribute exact zeros); per-lane partial sums like the per-sample terms ----
        if ((mask & ((1u << 10) | (1u << 11))) && lane < D && wv == 0) {
            const T e = a.X[cand * D + lane] - tk.xref[lane];
            const T e2 = e * e;
            if (lane < 3 * nf) t_via = (mask & (1u << 10)) ? e2 : T(0);
            else t_ydev = (mask & (1u << 11)) ? e2 : T(0);
        }
        if (FOLD) {
            part = mp_fma(wf[10], t_via, part);
            part = mp_fma(wf[11], t_ydev, part);
        }
        // goal approach alignment: v = P[S-1] - P[first] (telescoped sum of the tail diffs)
        gn_first = S - goal_tail_n;
        gn_first = gn_first < 0 ? 0 : gn_first;
        if ((mask & (1u << 14)) && ldu(&tk.has_normal) && S >= 3 && gn_first < S - 1) {
            T vx = RF(S - 1, kP0) - RF(gn_first, kP0), vy = RF(S - 1, kP1) - RF(gn_first, kP1), vz = RF(S - 1, kP2) - RF(gn_first, kP2);
            T n, inv_n;
            sqrt_rsqrt(vx * vx + vy * vy + vz * vz, n, inv_n);
            if (!(n < T(1e-12))) {
                T ux = vx * inv_n, uy = vy * inv_n, uz = vz * inv_n;
                if (ux != T(0) || uy != T(0) || uz != T(0)) {
                    const T sx = ldu(&tk.normal[0]), sy = ldu(&tk.normal[1]), sz = ldu(&tk.normal[2]);
                    T c = ux * -sx + uy * -sy + uz * -sz;
                    T cc = c < T(-1) ? T(-1) : (c > T(1) ? T(1) : c);
                    T e = T(1) - cc;
                    t_gn = e * e;
                    if (GRAD && c > T(-1) && c < T(1)) {
                        T us = ux * sx + uy * sy + uz * sz;
                        T k = w[14] * T(-2) * e;
                        gnx = k * -(sx - ux * us) * inv_n;
                        gny = k * -(sy - uy * us) * inv_n;
                        gnz = k * -(sz - uz * us) * inv_n;
                    }
                }
            }
        }

        }
        MP_PHASE(5);  // closed-form terms
        MP_ARGS_REFRESH();
        {
        MP_LOCALS;
        // ---- reductions over the wave and outputs ----
        T term[kNTerms];
        T jtot = T(0);
        if constexpr (WIDE) {
            // per-wave sum and minimum, then the waves' values combined in wave order (every thread ends up with the totals)
            const T ps = wave_sum(part), pm = wave_min(mc);
            if (lane == 0) { RED[wv * 2] = ps; RED[wv * 2 + 1] = pm; }
            __syncthreads();
            jtot = RED[0];
            mc = RED[1];
            for (int v = 1; v < NW; ++v) {
                jtot += RED[v * 2];
                mc = mp_min(mc, RED[v * 2 + 1]);
            }
            jtot = mp_fma(wf[14], t_gn, jtot);
        } else if (!COMPACT && a.terms) {
            term[0] = (mask & 1u) ? wave_sum(t_len) : T(0);
            term[1] = ((mask & 2u) && S >= 3) ? wave_sum(t_curv) * du : T(0);
            term[2] = ((mask & 4u) && S >= 3) ? wave_sum(t_ys) : T(0);
            term[3] = wave_sum(t_safe);
            term[4] = wave_sum(t_pref);
            term[5] = wave_sum(t_reb);
            term[6] = wave_sum(t_gc);
            term[7] = wave_sum(t_gt);
            term[8] = wave_sum(t_ac);
            term[9] = wave_sum(t_acol);
            term[10] = wave_sum(t_via);
            term[11] = wave_sum(t_ydev);
            term[12] = (mask & (1u << 12)) ? wave_sum(t_mono) : T(0);
            term[13] = (mask & (1u << 13)) ? wave_sum(t_sched) : T(0);
            term[14] = t_gn;
#pragma unroll
            for (int i = 0; i < kNTerms; ++i)
                if (mask & (1u << i)) jtot = (i == 0) ? w[i] * term[i] : jtot + w[i] * term[i];
        } else {
            // cost only: weight the per-lane partial sums first, one wave reduction instead of twelve.  The order of the
            // additions is the order in which the phases produce the sums, so that the five-wave instance, which folds
            // them phase by phase into one register pair, gives the same bits.
            if (!COMPACT) {
                // (the same fma per term as the folding instances, which take mask-free weights from the host: a set bit
                // means wf == w, and wf[1] = w_curv du rounded the same way)
                if (mask & 1u) part = mp_fma(w[0], t_len, part);
                if ((mask & 4u) && S >= 3) part = mp_fma(w[2], t_ys, part);
                if (mask & (1u << 12)) part = mp_fma(w[12], t_mono, part);
                if (mask & (1u << 13)) part = mp_fma(w[13], t_sched, part);
                if ((mask & 2u) && S >= 3) part = mp_fma(w[1] * du, t_curv, part);
                if (mask & (1u << 3)) part = mp_fma(w[3], t_safe, part);
                if (mask & (1u << 4)) part = mp_fma(w[4], t_pref, part);
                if (mask & (1u << 5)) part = mp_fma(w[5], t_reb, part);
                if (mask & (1u << 6)) part = mp_fma(w[6], t_gc, part);
                if (mask & (1u << 7)) part = mp_fma(w[7], t_gt, part);
                if (mask & (1u << 8)) part = mp_fma(w[8], t_ac, part);
                if (mask & (1u << 9)) part = mp_fma(w[9], t_acol, part);
                if (mask & (1u << 10)) part = mp_fma(w[10], t_via, part);
                if (mask & (1u << 11)) part = mp_fma(w[11], t_ydev, part);
            }
            jtot = wave_sum(part);
            if (COMPACT) jtot = mp_fma(wf[14], t_gn, jtot);
            else if (mask & (1u << 14)) jtot = mp_fma(w[14], t_gn, jtot);
        }
        if constexpr (!WIDE) mc = wave_min(mc);
        if (lane == 0 && wv == 0) {
            a.cost[cand] = jtot;
            if (a.minc) a.minc[cand] = mc;
            if (a.flags) a.flags[cand] = (uint8_t)((mc < T(0) ? 1u : 0u) | (mc >= req ? 2u : 0u));
        }
        if (!FOLD && a.terms && lane < kNTerms) {
            T v = T(0);
#pragma unroll
            for (int i = 0; i < kNTerms; ++i) v = (lane == i) ? term[i] : v;
            a.terms[cand * kNTerms + lane] = v;
        }

        }
        if (GRAD) {
            MP_PHASE(6);  // reductions + outputs
            MP_CAND_SYNC();
            MP_PHASE(7);  // grad: clearance chain
            MP_ARGS_REFRESH();
            {
            MP_LOCALS;
            // (2) curvature: T1 = A1 + G^T A2 (in place in A1*).  (G^T v)[s] = cm v[s-1] + c0 v[s] + cp v[s+1]
            // with the transposed np.gradient coefficients of row s (ends one-sided, inside central).
            const bool curv = (mask & 2u) && S >= 3;
            if (curv) {
                for (int s = lane + 64 * wv; s < S; s += 64 * NW) {
                    const T cm = s == 1 ? inv_du : ((s >= 2) ? inv_h2 : T(0));
                    const T cp = s == S - 2 ? -inv_du : ((s <= S - 3) ? -inv_h2 : T(0));
                    const T c0 = s == 0 ? -inv_du : (s == S - 1 ? inv_du : T(0));
                    const int lo = s == 0 ? 0 : s - 1, hi = s == S - 1 ? s : s + 1;
                    RF(s, kA1x) = mp_fma(cm, RF(lo, kA2x), mp_fma(c0, RF(s, kA2x), mp_fma(cp, RF(hi, kA2x), RF(s, kA1x))));
                    RF(s, kA1y) = mp_fma(cm, RF(lo, kA2y), mp_fma(c0, RF(s, kA2y), mp_fma(cp, RF(hi, kA2y), RF(s, kA1y))));
                    RF(s, kA1z) = mp_fma(cm, RF(lo, kA2z), mp_fma(c0, RF(s, kA2z), mp_fma(cp, RF(hi, kA2z), RF(s, kA1z))));
                }
            }
            MP_CAND_SYNC();
            }
            MP_PHASE(8);  // grad: curvature transpose
            MP_ARGS_REFRESH();
            {
            MP_LOCALS;
            const bool curv = (mask & 2u) && S >= 3;
            const T y0 = ldu(&tk.yaw_ctrl[0]), y1 = ldu(&tk.yaw_ctrl[Cy - 1]);
            // (3) gather every contribution to gP[s], gyaw[s]
            for (int s = lane + 64 * wv; s < S; s += 64 * NW) {
                T gx = T(0), gy = T(0), gz = T(0), gw = T(0);
                {   // clearance chain: d cost/d d_k * d d_k/d(P, yaw) of the collision sample taken at this spline sample
                    const int k = a.idx_identity ? s : a.inv_idx[s];
                    if (k >= 0 && k < Sp) {
                        const T gd = RF(k, kGD);
                        const int arg = FIVE ? (int)ARG16[k] : ARG[k];
                        if (gd != T(0) && arg >= 0) {
                            const BoxRec<T> bx = boxes[arg];
                            if (MODEL == 1) {
                                T g0, g1, g2, g3;
                                if (ROT == 2 && box_is_tilted(bx))
                                    tilt_grad(RF(s, kP0), RF(s, kP1), RF(s, kP2), RF(k, kCC), RF(k, kSS), a.ax, a.ay, a.az, bx, a.tilt, g0, g1, g2, g3);
                                else
                                    pair_grad(RF(s, kP0), RF(s, kP1), RF(s, kP2), RF(k, kCC), RF(k, kSS), a.ax, a.ay, a.az, bx, ROT != 0, g0, g1, g2, g3);
                                gx = gd * g0; gy = gd * g1; gz = gd * g2;
                                gw = gd * g3 * a.trig.deg2rad;
                            } else {
                                T g0, g1, g2;
                                point_grad(RF(s, kP0), RF(s, kP1), RF(s, kP2), bx, ROT != 0, ROT == 2 ? a.tilt : nullptr, g0, g1, g2);
                                gx = gd * g0; gy = gd * g1; gz = gd * g2;
                            }
                        }
                    }
                }
                const T px = RF(s, kP0), py = RF(s, kP1), pz = RF(s, kP2), yw = RF(s, kYW);
                if (mask & 1u) {  // RF(i, kSEGW) = w_len / |P[i+1] - P[i]| from the forward pass
                    if (s >= 1) {
                        const T k = RF(s - 1, kSEGW);
                        gx = mp_fma(k, px - RF(s - 1, kP0), gx); gy = mp_fma(k, py - RF(s - 1, kP1), gy); gz = mp_fma(k, pz - RF(s - 1, kP2), gz);
                    }
                    if (s + 1 < S) {
                        const T k = -RF(s, kSEGW);
                        gx = mp_fma(k, RF(s + 1, kP0) - px, gx); gy = mp_fma(k, RF(s + 1, kP1) - py, gy); gz = mp_fma(k, RF(s + 1, kP2) - pz, gz);
                    }
                }
                if (curv) {
                    const T cm = s == 1 ? inv_du : ((s >= 2) ? inv_h2 : T(0));
                    const T cp = s == S - 2 ? -inv_du : ((s <= S - 3) ? -inv_h2 : T(0));
                    const T c0 = s == 0 ? -inv_du : (s == S - 1 ? inv_du : T(0));
                    const int lo = s == 0 ? 0 : s - 1, hi = s == S - 1 ? s : s + 1;
                    gx = mp_fma(cm, RF(lo, kA1x), mp_fma(c0, RF(s, kA1x), mp_fma(cp, RF(hi, kA1x), gx)));
                    gy = mp_fma(cm, RF(lo, kA1y), mp_fma(c0, RF(s, kA1y), mp_fma(cp, RF(hi, kA1y), gy)));
                    gz = mp_fma(cm, RF(lo, kA1z), mp_fma(c0, RF(s, kA1z), mp_fma(cp, RF(hi, kA1z), gz)));
                }
                if (s == S - 1) { gx += gnx; gy += gny; gz += gnz; }
                if (s == gn_first) { gx -= gnx; gy -= gny; gz -= gnz; }
                if ((mask & 4u) && S >= 3) {  // yaw smoothness: e_i = y_i - 2 y_{i+1} + y_{i+2}
                    const T k2 = w2[2], k4 = T(-2) * k2;
                    if (s + 2 < S) gw = mp_fma(k2, mp_fma(T(-2), RF(s + 1, kYW), yw) + RF(s + 2, kYW), gw);
                    if (s >= 1 && s + 1 < S) gw = mp_fma(k4, mp_fma(T(-2), yw, RF(s - 1, kYW)) + RF(s + 1, kYW), gw);
                    if (s >= 2) gw = mp_fma(k2, mp_fma(T(-2), RF(s - 1, kYW), RF(s - 2, kYW)) + yw, gw);
                }
                if (mask & (1u << 12)) {
                    const int yaw_inc = ldu(&tk.yaw_inc);
                    const T ksg = yaw_inc ? -w2[12] : w2[12];
                    if (s >= 1) {
                        T dy = yw - RF(s - 1, kYW);
                        gw = mp_fma(ksg, npmax0(yaw_inc ? -dy : dy), gw);
                    }
                    if (s + 1 < S) {
                        T dy = RF(s + 1, kYW) - yw;
                        gw = mp_fma(-ksg, npmax0(yaw_inc ? -dy : dy), gw);
                    }
                }
                if (mask & (1u << 13)) {
                    gw = mp_fma(w2[13], yw - mp_fma(y1 - y0, tsched[s], y0), gw);
                }
                RF(s, kGX) = gx; RF(s, kGY) = gy; RF(s, kGZ) = gz; RF(s, kGW) = gw;   // slots dead by now (see RecField)
            }
            MP_CAND_SYNC();
            }
            MP_PHASE(9);  // grad: gather
            MP_ARGS_REFRESH();
            {
            MP_LOCALS;
            // (4) x-gradient: M^T g.  Eight outputs per pass: lane = (output o = lane / 8, sample group g = lane % 8),
            // every lane sums its output over the samples s = g, g + 8, ... and three DPP steps add the eight groups
            // (instead of one full wave reduction per output with all lanes on the same output).
            T *gout = a.grad + cand * D;
            const int og = lane >> 3, sg8 = lane & 7;
            for (int o0 = 0; o0 < D; o0 += 8) {
                const int o = o0 + og;
                T acc = T(0);
                if (o < D) {
                    const bool is_pos = o < 3 * nf;
                    const int jj = is_pos ? o / 3 : o - 3 * nf;
                    const int comp = is_pos ? o - 3 * jj : 0;
                    const T *V = REC + (is_pos ? (int)kGX + comp : (int)kGW);
                    const T *Mc = is_pos ? M + ff + jj : My + 1 + jj;
                    const int ld = is_pos ? C : Cy;
                    // eight (M, g) pairs fetched per trip, then the same in-order fma chain (a one-at-a-time loop
                    // waits for every load in turn)
                    for (int b0 = sg8 + 64 * wv; b0 < S; b0 += 64 * NW) {
                        T mv[8], gv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int sidx = b0 + 8 * u;
                            const int sc = sidx < S ? sidx : S - 1;
                            mv[u] = Mc[(size_t)sc * ld];
                            gv[u] = sidx < S ? V[sc * NREC] : T(0);
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc = mp_fma(mv[u], gv[u], acc);
                    }
                }
                acc += dpp_mov<0xB1, 0xf, true>(acc);   // quad_perm [1,0,3,2]
                acc += dpp_mov<0x4E, 0xf, true>(acc);   // quad_perm [2,3,0,1]
                acc += dpp_mov<0x141, 0xf, true>(acc);  // row_half_mirror: the other quad of this group of 8
                if constexpr (WIDE) {
                    if (o < D && sg8 == 0) RED[wv * RS + o] = acc;   // this wave's share of the samples
                } else if (o < D && sg8 == 0) {
                    const bool is_pos = o < 3 * nf;
                    if (mask & (1u << (is_pos ? 10 : 11))) acc = mp_fma(is_pos ? w2[10] : w2[11], a.X[cand * D + o] - tk.xref[o], acc);
                    gout[o] = acc;
                }
            }
            if constexpr (WIDE) {   // the waves' shares added in wave order, one output per thread (D <= 64)
                __syncthreads();
                const int o = lane + 64 * wv;
                if (o < D) {
                    T acc = RED[o];
                    for (int v = 1; v < NW; ++v) acc += RED[v * RS + o];
                    const bool is_pos = o < 3 * nf;
                    if (mask & (1u << (is_pos ? 10 : 11))) acc = mp_fma(is_pos ? w2[10] : w2[11], a.X[cand * D + o] - tk.xref[o], acc);
                    gout[o] = acc;
                }
            }
            }
        }
        MP_PHASE(10);  // grad: M^T g (or, cost only: reductions + outputs)
        MP_CAND_SYNC();  // the next candidate reuses the slice
    }
    MP_PHASE_FLUSH;
}
#undef RF
#undef a
#undef MP_ARGS_REFRESH
#undef MP_LOCALS
#undef MP_CAND_SYNC

// ---- split mode: the clearance profile on its own, one wave per (candidate, 64-sample chunk) --------------
// With many samples per trajectory the LDS slice of eval_kernel leaves room for very few waves per CU, while
// almost all of the work is the (sample x box) clearance phase.  This kernel needs ~5 KB of LDS per wave: it
// recomputes the chunk's spline samples with the same fma chain (same bits) and writes d and the deciding box.
__host__ __device__ inline int clearance_lds_bytes_per_wave(int e)
{
    return (64 + kChunkRec * 64) * e + 8 * 64 + 4 * 64 + 4 * 128;   // W | REC[64][7] | MINKEY | ARG | QUEUE
}

// ARGS: also write the deciding boxes (a gradient launch follows); cost-only launches skip their bookkeeping
template <typename T, int ROT, bool ARGS = true>
__global__ void __launch_bounds__(256, (ROT == 2 ? 2 : MP_CLR_WAVES)) clearance_kernel(const EvalArgs<T> a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    const int C = a.C, Cy = a.Cy, nf = a.nf, ff = a.ff, nyf = a.nyf;
    const int D = 3 * nf + nyf;
    unsigned char *slice = smem_raw + (size_t)wib * clearance_lds_bytes_per_wave((int)sizeof(T));
    T *W = reinterpret_cast<T *>(slice);
    T *REC = W + 64;
    unsigned long long *MINKEY = reinterpret_cast<unsigned long long *>(REC + kChunkRec * 64);
    int *ARG = reinterpret_cast<int *>(MINKEY + 64);
    unsigned int *QUEUE = reinterpret_cast<unsigned int *>(ARG + 64);
    const int chunks = (a.Sp + 63) >> 6;
    const long long units = a.n * chunks;

    for (long long u = (long long)blockIdx.x * wpb + wib; u < units; u += (long long)gridDim.x * wpb) {
        const long long cand = u / chunks;
        const int k0 = (int)(u - cand * chunks) << 6;
        long long tl, grp = 0;
        if (a.group_size > 0) {
            grp = cand / a.group_size;
            if (a.group_done && a.group_done[grp]) continue;
            tl = a.group_task ? (long long)a.group_task[grp] : 0;
        } else {
            tl = a.task_of ? (long long)a.task_of[cand] : (a.cpt > 0 ? cand / a.cpt : 0);
        }
        if (tl < 0 || tl >= (long long)a.n_tasks) continue;   // eval_kernel reports the candidate as NaN
        const int t = __builtin_amdgcn_readfirstlane((int)tl);
        const DevTask<T> &tk = a.tasks[t];
        const T *x = a.X + cand * D;
        const BoxRec<T> *boxes = a.boxes + ldu(&tk.box_begin);
        const BoxRec32 *boxes32 = a.boxes32 + ldu(&tk.sbox_begin);
        const BoxRec32 *groups = a.groups + 2 * ldu(&tk.group_begin);
        const int ngroups = ldu(&tk.group_count);
        const int n_wide = ldu(&tk.n_wide);
        const int k = k0 + lane;
        if (C == 4 && Cy == 4 && ff == 1 && nf == 2 && nyf == 2) {   // default layout: no waypoint table (eval_kernel, spline_sample4)
            T xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] = x[i];
            const TaskEnds<T> te = load_task_ends<T>(tk);
            if (k < a.Sp) {
                const int s = sample_of(a, k);
                T px, py, pz, yw;
                spline_sample4(a.M + (size_t)s * 4, a.My + (size_t)s * 4, te, xv, px, py, pz, yw);
                T *r = REC + lane * kChunkRec;
                r[kP0] = px; r[kP1] = py; r[kP2] = pz; r[kYW] = yw;
            }
        } else {
            {   // waypoint table, as in eval_kernel
                const int rows = C > Cy ? C : Cy;
                for (int i = lane; i < rows * 4; i += 64) {
                    const int j = i >> 2, col = i & 3;
                    T v = T(0);
                    if (col < 3) {
                        if (j < C) v = (j >= ff && j < ff + nf) ? x[3 * (j - ff) + col] : tk.ctrl[j][col];
                    } else if (j < Cy) {
                        v = (j >= 1 && j <= nyf) ? x[3 * nf + (j - 1)] : tk.yaw_ctrl[j];
                    }
                    W[i] = v;
                }
            }
            wave_lds_sync();
            if (k < a.Sp) {   // this lane's sample, same fma chain as eval_kernel
                const int s = sample_of(a, k);
                T px, py, pz, yw;
                spline_sample(a.M + (size_t)s * C, a.My + (size_t)s * Cy, C, Cy, W, px, py, pz, yw);
                T *r = REC + lane * kChunkRec;
                r[kP0] = px; r[kP1] = py; r[kP2] = pz; r[kYW] = yw;
            }
        }
        wave_lds_sync();
        int qcarry = 0;
        clearance_chunk<T, ROT, true, true, kChunkRec, 1, 1, false, ARGS, (ROT < 2)>(a, tk, boxes, boxes32, a.pairs + ldu(&tk.spair_begin), groups, ngroups,
                                                       n_wide, lane, k0, a.Sp, REC, MINKEY, ARG, QUEUE,
                                                       a.seed ? a.seed + (size_t)grp * a.Sp : nullptr, qcarry, false, false);
        if (k < a.Sp) {
            a.pre_d[cand * a.Sp + k] = key_to_distance<T>(MINKEY[lane]);
            if (ARGS) a.pre_arg[cand * a.Sp + k] = ARG[lane];
        }
        wave_lds_sync();  // the next unit reuses the slice
    }
}

// ---- stand-alone batched distance queries ------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) point_sdf_kernel(long long n, const T *__restrict__ pts,
                                                         const BoxRec<T> *__restrict__ boxes, int nb, T radius,
                                                         int has_rot, const T *__restrict__ tilt, T *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int arg;
        out[i] = point_distance(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], boxes, nb, radius, has_rot != 0, tilt, arg);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) box_sdf_kernel(long long n, const T *__restrict__ pos, const T *__restrict__ yaw,
                                                       const BoxRec<T> *__restrict__ boxes, int nb, T ax, T ay, T az,
                                                       const T *__restrict__ tilt, T *__restrict__ out, const TrigTab<T> trig)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        T c, s;
        yaw_rot(yaw[i], trig, c, s);
        Pose<T> ps;
        pose_setup(ps, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], c, s, ax, ay);
        int arg;
        out[i] = pose_distance(ps, ax, ay, az, boxes, nb, tilt, arg);
    }
}


// ---- device-resident cross-entropy method (many instances in lock-step) ------------------
// Restates _cem_optimize (geom/spline_opt.py:222-277) per instance: sample a population around
// (mu, sigma), evaluate it with eval_kernel, keep the best candidate seen, refit (mu, sigma) to the
// elite_count best members with alpha-smoothing, stop on |best - prev_best| < tol and max sigma < 0.03.
struct Philox {
    // Philox-4x32-10 counter-based generator (Salmon et al., SC'11): stateless, one call per 4 words
    __device__ static void round(unsigned int (&c)[4], unsigned int k0, unsigned int k1)
    {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned int h0 = (unsigned int)(p0 >> 32), l0 = (unsigned int)p0;
        const unsigned int h1 = (unsigned int)(p1 >> 32), l1 = (unsigned int)p1;
        c[0] = h1 ^ c[1] ^ k0; c[1] = l1; c[2] = h0 ^ c[3] ^ k1; c[3] = l0;
    }
    __device__ static void generate(unsigned long long seed, unsigned long long ctr_lo, unsigned long long ctr_hi,
                                    unsigned int (&out)[4])
    {
        unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
        out[0] = (unsigned int)ctr_lo; out[1] = (unsigned int)(ctr_lo >> 32);
        out[2] = (unsigned int)ctr_hi; out[3] = (unsigned int)(ctr_hi >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            round(out, k0, k1);
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
    }
};

template <typename T> struct CemState {
    T *mu, *sigma, *best_x;      // [n_inst x D]
    T *best_val, *prev_best;     // [n_inst]
    T *best_minc;                // [n_inst]
    uint8_t *best_flags, *done;  // [n_inst]
    int *nit;                    // [n_inst]
    int *n_active;               // [1] instances still running after this generation
    int *gen;                    // [1] generation counter (device resident, so that a captured graph can be replayed)
};

// X[i, p, :] = mu[i] + eps * sigma[i]; eps from `noise` (generation-major, parity mode) or Philox + Box-Muller
template <typename T>
__global__ void __launch_bounds__(256) cem_sample_kernel(long long n_inst, int pop, int D, CemState<T> st,
                                                          const T *__restrict__ noise, unsigned long long seed,
                                                          unsigned long long inst_base, T *__restrict__ X)
{
    const long long total = n_inst * pop * D;
    const int gen = *st.gen;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long inst = e / ((long long)pop * D);
        if (st.done[inst]) continue;
        const int d = (int)(e % D);
        T eps;
        if (noise) {
            eps = noise[e];
        } else {
            unsigned int r[4];
            // the counter is the element's index in the GLOBAL list of instances (inst_base = this call's first one)
            Philox::generate(seed, inst_base * (unsigned long long)pop * D + (unsigned long long)e, (unsigned long long)gen, r);
            const double u1 = ((double)r[0] * 4294967296.0 + (double)r[1] + 1.0) * (1.0 / 18446744073709551616.0);  // (0,1]
            const double u2 = ((double)r[2] * 4294967296.0 + (double)r[3]) * (1.0 / 18446744073709551616.0);        // [0,1)
            eps = (T)(::sqrt(-2.0 * ::log(u1)) * ::cospi(2.0 * u2));
        }
        X[e] = st.mu[inst * D + d] + eps * st.sigma[inst * D + d];
    }
}

// between evaluation and update of a generation: advance the generation counter, clear the running count
template <typename T> __global__ void cem_tick_kernel(CemState<T> st)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { *st.gen += 1; *st.n_active = 0; }
}

// one wave per instance: best-so-far, elite refit, stop rule
template <typename T>
__global__ void __launch_bounds__(256) cem_update_kernel(long long n_inst, int pop, int D, int n_elite, T alpha, T min_sigma,
                                                          T tol, CemState<T> st, const T *__restrict__ X,
                                                          const T *__restrict__ vals, const T *__restrict__ minc,
                                                          const uint8_t *__restrict__ flags)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wpb = (int)(blockDim.x >> 6);
    T *V = reinterpret_cast<T *>(smem_raw) + (size_t)wib * pop;   // this wave's copy of the population costs
    // instances still running are counted per block and added to the device counter ONCE per block: one atomic per instance
    // on that single address (128,000 of them at config 5) took 1.07 of the kernel's 1.08 ms - a device-scope atomic on one
    // address completes every ~12 ns on this multi-XCD part
    __shared__ int block_active;
    if (threadIdx.x == 0) block_active = 0;
    __syncthreads();
    int wave_active = 0;
    for (long long inst = (long long)blockIdx.x * wpb + wib; inst < n_inst; inst += (long long)gridDim.x * wpb) {
        if (st.done[inst]) continue;
        const T *v = vals + inst * pop;
        const T *x = X + inst * (long long)pop * D;
        // argmin (first minimum, NaN never wins) and LDS copy
        T bv = Lim<T>::inf();
        int bi = 0x7fffffff;
        for (int p = lane; p < pop; p += 64) {
            T val = v[p];
            val = (val == val) ? val : Lim<T>::inf();
            V[p] = val;
            if (val < bv) { bv = val; bi = p; }
        }
        wave_lds_sync();
        const T wmin = wave_min(bv);
        int cand_i = (bv == wmin) ? bi : 0x7fffffff;
        const int best_i = wave_min(cand_i);
        T best_val = st.best_val[inst];
        if (best_i < pop && wmin < best_val) {
            best_val = wmin;
            for (int d = lane; d < D; d += 64) st.best_x[inst * D + d] = x[(long long)best_i * D + d];
            if (lane == 0) {
                st.best_val[inst] = wmin;
                st.best_minc[inst] = minc[inst * pop + best_i];
                st.best_flags[inst] = flags[inst * pop + best_i];
            }
        }
        // elite membership by rank (ties broken by index, like a stable partition)
        unsigned int elite_bits = 0;  // bit j: member lane + 64 j is an elite
        for (int p = lane, j = 0; p < pop; p += 64, ++j) {
            const T mine = V[p];
            int rank = 0;
            for (int q = 0; q < pop; ++q) {
                const T other = V[q];
                rank += (other < mine || (other == mine && q < p)) ? 1 : 0;
            }
            if (rank < n_elite) elite_bits |= 1u << j;
        }
        // refit per dimension: mean, then population std (np.std, ddof = 0).  Eight dimensions at a time: their loads are
        // issued together and their wave sums are independent (one dimension after the other, the kernel waited for sixteen
        // load round trips and sixteen dependent reductions per instance); every sum adds in the same order
        // as before (per lane over its members, then across the wave), so the bits are unchanged.
        T max_sigma = T(0);
        for (int d0 = 0; d0 < D; d0 += 8) {
            const int nd = D - d0 < 8 ? D - d0 : 8;
            T acc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] = T(0);
            for (int p = lane, j = 0; p < pop; p += 64, ++j)
                if (elite_bits & (1u << j)) {
                    const T *row = x + (long long)p * D + d0;
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (u < nd) acc[u] += row[u];
                }
            T mean[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) mean[u] = u < nd ? wave_sum(acc[u]) / (T)n_elite : T(0);
            T acc2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc2[u] = T(0);
            for (int p = lane, j = 0; p < pop; p += 64, ++j)
                if (elite_bits & (1u << j)) {
                    const T *row = x + (long long)p * D + d0;
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (u < nd) {
                            const T e = row[u] - mean[u];
                            acc2[u] += e * e;
                        }
                }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u < nd) {
                    const int d = d0 + u;
                    const T sd = mp_sqrt(wave_sum(acc2[u]) / (T)n_elite) + min_sigma;
                    const T mu_new = alpha * st.mu[inst * D + d] + (T(1) - alpha) * mean[u];
                    T sg_new = alpha * st.sigma[inst * D + d] + (T(1) - alpha) * sd;
                    sg_new = sg_new > min_sigma ? sg_new : min_sigma;
                    max_sigma = sg_new > max_sigma ? sg_new : max_sigma;
                    if (lane == 0) { st.mu[inst * D + d] = mu_new; st.sigma[inst * D + d] = sg_new; }
                }
        }
        if (lane == 0) {
            const T prev = st.prev_best[inst];
            st.nit[inst] += 1;
            const T diff = prev - best_val;
            const bool stop = ((diff < T(0) ? -diff : diff) < tol) && (max_sigma < T(0.03));
            if (stop) st.done[inst] = 1;
            else wave_active += 1;
            st.prev_best[inst] = best_val;
        }
        wave_lds_sync();
    }
    if (lane == 0 && wave_active) atomicAdd(&block_active, wave_active);
    __syncthreads();
    if (threadIdx.x == 0 && block_active) atomicAdd(st.n_active, block_active);
}

template <typename T>
__global__ void cem_init_kernel(long long n_inst, int D, CemState<T> st, const T *__restrict__ x0, const T *__restrict__ sigma0,
                                T min_sigma)
{
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n_inst * D; e += (long long)gridDim.x * blockDim.x) {
        st.mu[e] = x0[e];
        st.best_x[e] = x0[e];
        const T sg = sigma0[e];
        st.sigma[e] = sg > min_sigma ? sg : min_sigma;
        if (e < n_inst) {
            st.best_val[e] = Lim<T>::inf();
            st.prev_best[e] = Lim<T>::inf();
            st.best_minc[e] = Lim<T>::inf();
            st.best_flags[e] = 0;
            st.done[e] = 0;
            st.nit[e] = 0;
        }
    }
}

}  // namespace mp
