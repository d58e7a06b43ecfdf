/*
 * Plain-C sharded job over the C ABI with RCCL (no Python, no torch): the multi-GPU step of SURVEY.md 8(e) for a caller
 * that brings its own communicator.  One process per rank:
 *
 *     cabi_gather <rank> <n_ranks> <id_file> [device]
 *
 * Rank 0 writes the ncclUniqueId to <id_file>, the others wait for it; rank r evaluates candidates
 * [r * N, (r + 1) * N) of a job whose candidates are a function of their GLOBAL index on GPU `device` (default: r),
 * packs nothing — cost | min_clear are adjacent device arrays — and calls mp_allgather_results once.  Every rank then
 * re-evaluates ALL candidates locally and compares the gathered bytes with its own: prints "rank r: gathered == local"
 * and returns 0, or returns 1.  tests/test_gpu_evidence.py runs it with one rank on the one-GPU box (the 2..8-rank
 * form needs that many GPUs: `for r in 0 1; do ./cabi_gather $r 2 /tmp/id & done`).
 *
 *   hipcc tests/cabi/cabi_gather.c -I include -L motion_planning_amd -lmp_hip -lrccl ...   (__graft_entry__.build)
 */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "mp_api.h"

#define S 32
#define C4 4
#define N 512 /* candidates per rank */
#define D 8

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t e_ = (x); if (e_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(e_)); return 2; } } while (0)
#define CHECK_MP(x) do { int e_ = (x); if (e_) { fprintf(stderr, "%s: %s\n", #x, mp_last_error(ctx)); return 2; } } while (0)

/* candidate g of the job: a deterministic spread around the straight line */
static void candidate(long g, double *x)
{
    const double base[D] = {-0.4, -0.2, 0.9, 0.2, 0.1, 0.8, 20.0, 40.0};
    unsigned long long s = 0x9E3779B97F4A7C15ull * (unsigned long long)(g + 1);
    for (int d = 0; d < D; ++d) {
        s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ull; s ^= s >> 32;
        const double u = (double)(s >> 11) / 9007199254740992.0 - 0.5;
        x[d] = base[d] + (d < 6 ? 0.8 : 30.0) * u;
    }
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s rank n_ranks id_file [device]\n", argv[0]); return 2; }
    const int rank = atoi(argv[1]), nr = atoi(argv[2]);
    const int device = argc > 4 ? atoi(argv[4]) : rank;
    if (rank < 0 || nr < 1 || rank >= nr || nr > 64) { fprintf(stderr, "bad rank / n_ranks\n"); return 2; }
    CHECK_HIP(hipSetDevice(device));

    ncclUniqueId id;
    if (rank == 0) {
        CHECK_NCCL(ncclGetUniqueId(&id));
        char tmp[1024];
        snprintf(tmp, sizeof tmp, "%s.tmp", argv[3]);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(&id, sizeof id, 1, f) != 1) { fprintf(stderr, "cannot write %s\n", tmp); return 2; }
        fclose(f);
        if (rename(tmp, argv[3])) { fprintf(stderr, "cannot rename %s\n", tmp); return 2; }
    } else {
        FILE *f = NULL;
        for (int tries = 0; tries < 600 && !(f = fopen(argv[3], "rb")); ++tries) usleep(100000);
        if (!f || fread(&id, sizeof id, 1, f) != 1) { fprintf(stderr, "rank %d: no id in %s\n", rank, argv[3]); return 2; }
        fclose(f);
    }
    ncclComm_t comm;
    CHECK_NCCL(ncclCommInitRank(&comm, nr, id, rank));

    mp_ctx *ctx = NULL;
    if (mp_create(device, MP_F64, &ctx)) { fprintf(stderr, "mp_create: %s\n", mp_last_error(NULL)); return 2; }
    const double cx[3] = {0.0, 0.6, -0.3}, cy[3] = {0.0, 0.1, -0.2}, cz[3] = {0.05, 0.5, 0.7};
    const double hx[3] = {2.0, 0.2, 0.15}, hy[3] = {2.0, 0.3, 0.2}, hz[3] = {0.05, 0.4, 0.1};
    CHECK_MP(mp_set_scene(ctx, 3, cx, cy, cz, hx, hy, hz, NULL));
    static int32_t idx[S];
    static uint8_t mask[S];
    static double tsched[S];
    for (int i = 0; i < S; ++i) { idx[i] = i; mask[i] = (uint8_t)(i < S - 3); tsched[i] = (double)i / (S - 1); }
    mp_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_samples = S; cfg.n_ctrl = C4; cfg.first_free = 1; cfg.n_free = 2; cfg.n_yaw_ctrl = C4; cfg.n_yaw_free = 2;
    cfg.model = MP_MODEL_BOX; cfg.n_safe = S; cfg.keep_n = S; cfg.use_pref = 1; cfg.tail_start = S - 8;
    cfg.goal_tail_n = 4; cfg.term_mask = 0x7fff;
    for (int t = 0; t < MP_NTERMS; ++t) cfg.weights[t] = 1.0 + 0.5 * t;
    cfg.required_clearance = 0.0; cfg.preferred_clearance = 0.05; cfg.approach_target = 0.03; cfg.point_radius = 1e-6;
    cfg.half_extents[0] = 0.15; cfg.half_extents[1] = 0.1; cfg.half_extents[2] = 0.1;
    cfg.idx_safe = idx; cfg.approach_mask = mask; cfg.t_sched = tsched;
    mp_task task;
    memset(&task, 0, sizeof task);
    const double wp[C4][3] = {{-1.0, -0.5, 1.0}, {-0.4, -0.2, 0.9}, {0.2, 0.1, 0.8}, {1.0, 0.4, 0.3}};
    const double yaw[C4] = {0.0, 20.0, 40.0, 60.0};
    for (int j = 0; j < C4; ++j) { for (int a = 0; a < 3; ++a) task.ctrl[j][a] = wp[j][a]; task.yaw_ctrl[j] = yaw[j]; }
    task.goal_normal[2] = 1.0; task.has_goal_normal = 1; task.yaw_increasing = 1; task.box_count = 3;
    CHECK_MP(mp_set_problem(ctx, &cfg, 1, &task));

    /* all candidates of the job on the host (every rank can build any of them), this rank's shard on the device */
    const long total = (long)N * nr;
    double *X = (double *)malloc((size_t)total * D * sizeof(double));
    for (long g = 0; g < total; ++g) candidate(g, X + g * D);
    double *dX, *dRes, *dAll, *dXall, *dResAll;
    CHECK_HIP(hipMalloc((void **)&dX, (size_t)N * D * 8));
    CHECK_HIP(hipMalloc((void **)&dRes, (size_t)N * 2 * 8));              /* cost[N] | min_clear[N] */
    CHECK_HIP(hipMalloc((void **)&dAll, (size_t)total * 2 * 8));          /* rank r's pair of arrays at r * N * 2 */
    CHECK_HIP(hipMalloc((void **)&dXall, (size_t)total * D * 8));
    CHECK_HIP(hipMalloc((void **)&dResAll, (size_t)total * 2 * 8));
    CHECK_HIP(hipMemcpy(dX, X + (long)rank * N * D, (size_t)N * D * 8, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dXall, X, (size_t)total * D * 8, hipMemcpyHostToDevice));

    CHECK_MP(mp_eval(ctx, N, dX, NULL, dRes, NULL, NULL, dRes + N, NULL, NULL, MP_MEM_DEVICE));
    CHECK_MP(mp_allgather_results(ctx, comm, dRes, dAll, (size_t)N * 2 * 8));   /* behind the kernel, same stream */
    CHECK_MP(mp_synchronize(ctx));

    /* what one GPU computes for the whole job */
    CHECK_MP(mp_eval(ctx, total, dXall, NULL, dResAll, NULL, NULL, dResAll + total, NULL, NULL, MP_MEM_DEVICE));
    CHECK_MP(mp_synchronize(ctx));
    double *all = (double *)malloc((size_t)total * 2 * 8), *loc = (double *)malloc((size_t)total * 2 * 8);
    CHECK_HIP(hipMemcpy(all, dAll, (size_t)total * 2 * 8, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(loc, dResAll, (size_t)total * 2 * 8, hipMemcpyDeviceToHost));
    long bad = 0;
    for (int r = 0; r < nr; ++r)
        for (int i = 0; i < N; ++i) {
            const long g = (long)r * N + i;
            if (memcmp(&all[(long)r * 2 * N + i], &loc[g], 8)) ++bad;                 /* cost */
            if (memcmp(&all[(long)r * 2 * N + N + i], &loc[total + g], 8)) ++bad;     /* min clearance */
        }
    if (bad) { fprintf(stderr, "rank %d: %ld gathered values differ from the local evaluation\n", rank, bad); return 1; }
    printf("rank %d of %d: gathered == local (%ld candidates, cost[0] = %.17g)\n", rank, nr, total, all[0]);

    /* error path: a NULL communicator is refused */
    printf("null_comm rc=%d\n", mp_allgather_results(ctx, NULL, dRes, dAll, 8));
    mp_destroy(ctx);
    ncclCommDestroy(comm);
    return 0;
}
