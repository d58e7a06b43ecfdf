// mp_api.hip — C ABI of libmp_hip.so (see include/mp_api.h).  Host side: context, device
// buffers, problem upload, kernel launches.  There is no CPU compute path in this file:
// without a HIP device mp_create fails and nothing else can be called.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/mp_api.h"
#include "mp_kernels.hpp"

// ---- mp_eval_noalloc: the split-mode tables in a file-scope device array ------------------------------------------
// `scratch` is a host-side stand-in for the symbol, NOT a device address: host code only ever uses the per-device
// address that hipGetSymbolAddress returned in mp_create (g_scratch_ptr); kernels get that pointer as an argument.
static __device__ __align__(256) unsigned char scratch[MP_NOALLOC_SCRATCH_BYTES];

#ifndef MP_WPB_TWO
#define MP_WPB_TWO 1   // 0: A/B of the two-wave-block rule in launch_eval
#endif
#ifndef MP_WPB_PCT
#define MP_WPB_PCT 75   // a smaller block must put a third more waves on the CU to be preferred (measured: 85, 75, 60)
#endif
namespace {

std::atomic<void *> g_scratch_ptr[64];   // [device] address of that GPU's copy of `scratch` (each GPU has its own)
// Every context of a GPU shares that GPU's array.  Calls on ONE stream are ordered by the stream; a call that comes on
// another stream first waits (hipStreamWaitEvent: no allocation) for the event recorded behind the previous call, so two
// contexts or streams cannot race on the tables.  Calls made while a stream is being captured cannot take part (an event of
// uncaptured work must not enter a graph): replays of a captured call are ordered by the caller, as documented in mp_api.h.
struct ScratchOrder {
    std::mutex m;
    hipEvent_t done = nullptr;     // created by the first mp_create on the device
    hipStream_t last = nullptr;
    bool recorded = false;
};
ScratchOrder g_scratch_order[64];

thread_local char g_create_err[512] = "";

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

}  // namespace

struct mp_ctx {
    int device = 0;
    int dtype = MP_F64;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    char err[512] = "";
    int num_cu = 256;
    size_t max_lds = 65536;

    // scene
    DevBuf boxes, boxes32, pairs32, groups;
    DevBuf grid;                  // seed grids of the box ranges (problem data, built by upload_problem)
    int grid_mode = 1;            // 0: no seed grids (MP_GRID=0, tuning aid / A-B)
    int group_min = 0;     // > 0: box ranges from this size on are sorted along a Morton curve and grouped (MP_GROUP_MIN; 0: rule in upload_problem)
    int group_small = 4, group_large = 8;   // boxes per group of the bound tables below / from 1,000 boxes per range (MP_GROUP_SMALL / MP_GROUP_LARGE, even)
    int grid_cells_per_box = 16;   // resolution of a seed grid: about this many cells per box of the range (MP_GRID_CELLS)
    std::vector<double> h_boxes;  // host copy of the scene (n x 6) for the per-task bound tables
    float scene_radius = 0.f;
    int has_rot = 0;              // 0 axis-aligned scene, 1 some boxes have a yaw of their own, 2 some are tilted
    DevBuf tilt;                  // rotation matrices (9 values each, context dtype) of the tilted boxes
    int n_boxes = 0;
    bool have_scene = false;

    // problem
    bool have_problem = false;
    mp_config cfg{};
    int n_tasks = 0;
    long long cands_per_task = 0;
    std::vector<double> basis, basis_yaw;
    DevBuf d_idx, d_inv, d_mask, d_M, d_My, d_tsched, d_tasks;

    // staging for MP_MEM_HOST calls
    DevBuf s_X, s_taskof, s_cost, s_terms, s_grad, s_minc, s_flags, s_dist, s_aux0, s_aux1;
    void *h_pin = nullptr;        // pinned, device-visible host staging for small host-buffer calls (zero copy)
    size_t h_pin_cap = 0;
    DevBuf s_pre_d, s_pre_arg;    // split mode: clearance profile + deciding box of every (candidate, sample)
    DevBuf d_seed, d_seed_d;      // CEM: deciding boxes [n_inst x n_safe] of the instances' mean trajectories, and their clearances (cem_solve)
    int seed_mode = 1;            // 0: no seeds (MP_SEEDS=0, tuning aid / A-B)
    int defer_mode = 1;           // 0: every chunk of a trajectory drains its own queue (MP_DEFER=0, tuning aid / A-B)
    int max_task_boxes = 0;
    int idx_identity = 0;         // idx_safe is 0, 1, 2, ...
    int split_mode = -1;          // -1 automatic, 0 never, 1 always (MP_SPLIT_CLEARANCE, tuning aid)
    int wide_min_s = 176;         // the wide instance serves split-mode gradient launches with more samples than this (MP_WIDE_MIN_S)
    int wide_mode = -1;           // -1 automatic, 0 never (MP_WIDE=0, tuning aid / A-B of the wide split-mode instance)
    int five_wave_mode = -1;      // -1 automatic, 0 never (MP_FIVE_WAVES=0, tuning aid / A-B of the five-wave gradient instance)
    int cem_graph = 1;            // replay the CEM generations from a captured HIP graph (MP_CEM_GRAPH=0 turns it off)

    // device-resident CEM
    DevBuf c_X, c_vals, c_minc, c_flags, c_mu, c_sigma, c_bestx, c_bestval, c_prev, c_bminc, c_bflags, c_done, c_nit,
        c_active, c_task, c_x0, c_s0, c_noise;
};

namespace {

int fail(mp_ctx *ctx, int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx ? ctx->err : g_create_err, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail((ctx), MP_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int reserve(mp_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return MP_OK;
    {   // growing a buffer while the caller's stream is being captured into a graph would invalidate the capture
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (ctx->stream && hipStreamIsCapturing(ctx->stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        if (cap != hipStreamCaptureStatusNone)
            return fail(ctx, MP_E_STATE, "a device buffer would have to grow to %zu bytes while the stream is being captured: "
                        "make one ordinary call of this shape before capturing", bytes);
    }
    if (b.p) HIP_TRY(ctx, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = std::max<size_t>(bytes, 256);
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return MP_OK;
}

int upload(mp_ctx *ctx, DevBuf &b, const void *src, size_t bytes)
{
    int rc = reserve(ctx, b, bytes);
    if (rc) return rc;
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return MP_OK;
}

size_t esize(const mp_ctx *ctx) { return ctx->dtype == MP_F64 ? 8 : 4; }

// layout of the static scratch array: [n x S'] clearances, then (256-byte aligned) [n x S'] deciding boxes
size_t noalloc_arg_offset(size_t d_bytes) { return (d_bytes + 255) & ~(size_t)255; }

constexpr size_t kPinnedCallBytes = (size_t)256 << 10;  // host-buffer calls up to this size go through pinned memory

// ---- interpolating B-spline sampling matrix (host, once per problem) --------------------
// Restates what scipy.interpolate.make_interp_spline(u, y, k) with bc_type=None does for
// u = linspace(0,1,n) (not-a-knot knots), as a constant matrix: P = M @ W.
void linspace01(int n, std::vector<double> &out)
{
    out.resize(n);
    const double step = 1.0 / (double)(n - 1);
    for (int i = 0; i < n; ++i) out[i] = (double)i * step;
    out[n - 1] = 1.0;
}

void basis_funs(const std::vector<double> &t, int k, int n, double x, int &first, long double N[4])
{
    // span: t[i] <= x < t[i+1], clamped to the last non-empty interval
    int i = k;
    while (i < n - 1 && x >= t[i + 1]) ++i;
    long double left[4], right[4];
    N[0] = 1.0L;
    for (int j = 1; j <= k; ++j) {
        left[j] = (long double)x - (long double)t[i + 1 - j];
        right[j] = (long double)t[i + j] - (long double)x;
        long double saved = 0.0L;
        for (int r = 0; r < j; ++r) {
            long double tmp = N[r] / (right[r + 1] + left[j - r]);
            N[r] = saved + right[r + 1] * tmp;
            saved = left[j - r] * tmp;
        }
        N[j] = saved;
    }
    first = i - k;
}

int build_basis(int n, int k, int S, double *M)
{
    if (k < 1 || k > 3 || n < k + 1 || n > MP_MAX_CTRL || S < 2) return MP_E_ARG;
    std::vector<double> u, us, t;
    linspace01(n, u);
    linspace01(S, us);
    for (int i = 0; i <= k; ++i) t.push_back(u[0]);
    if (k % 2 == 1) {
        const int k2 = (k + 1) / 2;
        for (int i = k2; i < n - k2; ++i) t.push_back(u[i]);
    } else {
        for (int i = 1; i + 2 < n; ++i) t.push_back(0.5 * (u[i] + u[i + 1]));
    }
    for (int i = 0; i <= k; ++i) t.push_back(u[n - 1]);
    if ((int)t.size() != n + k + 1) return MP_E_ARG;

    // collocation matrix and its inverse (Gauss-Jordan, partial pivoting, long double)
    std::vector<long double> A((size_t)n * n, 0.0L), Inv((size_t)n * n, 0.0L);
    for (int r = 0; r < n; ++r) {
        int first;
        long double N[4];
        basis_funs(t, k, n, u[r], first, N);
        for (int j = 0; j <= k; ++j) A[(size_t)r * n + first + j] = N[j];
        Inv[(size_t)r * n + r] = 1.0L;
    }
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r)
            if (fabsl(A[(size_t)r * n + c]) > fabsl(A[(size_t)piv * n + c])) piv = r;
        if (A[(size_t)piv * n + c] == 0.0L) return MP_E_ARG;
        if (piv != c)
            for (int j = 0; j < n; ++j) {
                std::swap(A[(size_t)piv * n + j], A[(size_t)c * n + j]);
                std::swap(Inv[(size_t)piv * n + j], Inv[(size_t)c * n + j]);
            }
        long double d = A[(size_t)c * n + c];
        for (int j = 0; j < n; ++j) { A[(size_t)c * n + j] /= d; Inv[(size_t)c * n + j] /= d; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            long double f = A[(size_t)r * n + c];
            if (f == 0.0L) continue;
            for (int j = 0; j < n; ++j) {
                A[(size_t)r * n + j] -= f * A[(size_t)c * n + j];
                Inv[(size_t)r * n + j] -= f * Inv[(size_t)c * n + j];
            }
        }
    }
    for (int s = 0; s < S; ++s) {
        int first;
        long double N[4];
        basis_funs(t, k, n, us[s], first, N);
        for (int j = 0; j < n; ++j) {
            long double acc = 0.0L;
            for (int q = 0; q <= k; ++q) acc += N[q] * Inv[(size_t)(first + q) * n + j];
            M[(size_t)s * n + j] = (double)acc;
        }
    }
    return MP_OK;
}

template <typename T> void to_vec(const double *src, size_t n, std::vector<T> &out)
{
    out.resize(n);
    for (size_t i = 0; i < n; ++i) out[i] = (T)src[i];
}

// cb/sb: cos / sin of each box's own yaw (1, 0 = axis aligned); sb = 2 marks a tilted box whose cb is the index of its
// rotation matrix (world <- box, row major) in rtab
template <typename T> int upload_scene(mp_ctx *ctx, int n, const double *cx, const double *cy, const double *cz,
                                        const double *hx, const double *hy, const double *hz, const double *cb,
                                        const double *sb, const std::vector<double> &rtab)
{
    std::vector<mp::BoxRec<T>> rec((size_t)std::max(n, 1));
    int rot = 0;
    for (int i = 0; i < n; ++i) {
        rec[i].cx = (T)cx[i]; rec[i].cy = (T)cy[i]; rec[i].cz = (T)cz[i];
        rec[i].hx = (T)hx[i]; rec[i].hy = (T)hy[i]; rec[i].hz = (T)hz[i];
        rec[i].cb = (T)cb[i]; rec[i].sb = (T)sb[i];
        if (sb[i] == 2.0) rot = 2;
        else if (!(cb[i] == 1.0 && sb[i] == 0.0)) rot = std::max(rot, 1);
    }
    int rc = upload(ctx, ctx->boxes, rec.data(), (size_t)n * sizeof(mp::BoxRec<T>));
    if (rc) return rc;
    std::vector<T> rt;
    to_vec<T>(rtab.data(), rtab.size(), rt);
    if (rt.empty()) rt.assign(9, T(0));
    if ((rc = upload(ctx, ctx->tilt, rt.data(), rt.size() * sizeof(T)))) return rc;
    ctx->has_rot = rot;
    // host copy (centre + half extents of each box's WORLD AABB) + scene radius (scales the slack of the fp32
    // bound pass); the fp32 tables are built per task.  The AABB of a rotated box bounds the box, so the
    // bound pass stays a valid lower bound of the distance.
    ctx->h_boxes.resize((size_t)n * 6);
    double radius = 0.0;
    for (int i = 0; i < n; ++i) {
        double *r = &ctx->h_boxes[(size_t)i * 6];
        double hxw = hx[i], hyw = hy[i], hzw = hz[i];
        if (sb[i] == 2.0) {
            const double *R = &rtab[(size_t)9 * (size_t)cb[i]];
            hxw = (std::fabs(R[0]) * hx[i] + std::fabs(R[1]) * hy[i] + std::fabs(R[2]) * hz[i]) * (1.0 + 1e-12);
            hyw = (std::fabs(R[3]) * hx[i] + std::fabs(R[4]) * hy[i] + std::fabs(R[5]) * hz[i]) * (1.0 + 1e-12);
            hzw = (std::fabs(R[6]) * hx[i] + std::fabs(R[7]) * hy[i] + std::fabs(R[8]) * hz[i]) * (1.0 + 1e-12);
        } else if (!(cb[i] == 1.0 && sb[i] == 0.0)) {
            const double ac = std::fabs(cb[i]), as = std::fabs(sb[i]);
            hxw = (ac * hx[i] + as * hy[i]) * (1.0 + 1e-12);
            hyw = (as * hx[i] + ac * hy[i]) * (1.0 + 1e-12);
        }
        r[0] = cx[i]; r[1] = cy[i]; r[2] = cz[i]; r[3] = hxw; r[4] = hyw; r[5] = hzw;
        radius = std::max({radius, std::fabs(cx[i]) + hxw, std::fabs(cy[i]) + hyw, std::fabs(cz[i]) + hzw});
    }
    ctx->scene_radius = (float)radius;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // rec and rt are locals
    return MP_OK;
}

template <typename T> int upload_problem(mp_ctx *ctx, const mp_task *tasks)
{
    const mp_config &c = ctx->cfg;
    std::vector<T> v;
    to_vec<T>(ctx->basis.data(), ctx->basis.size(), v);
    int rc = upload(ctx, ctx->d_M, v.data(), v.size() * sizeof(T));
    if (rc) return rc;
    std::vector<T> vy;
    to_vec<T>(ctx->basis_yaw.data(), ctx->basis_yaw.size(), vy);
    if ((rc = upload(ctx, ctx->d_My, vy.data(), vy.size() * sizeof(T)))) return rc;
    std::vector<T> ts;
    to_vec<T>(c.t_sched, (size_t)c.n_samples, ts);
    if ((rc = upload(ctx, ctx->d_tsched, ts.data(), ts.size() * sizeof(T)))) return rc;
    std::vector<mp::DevTask<T>> dt((size_t)ctx->n_tasks);
    for (int i = 0; i < ctx->n_tasks; ++i) {
        for (int j = 0; j < MP_MAX_CTRL; ++j) {
            for (int a = 0; a < 3; ++a) dt[i].ctrl[j][a] = (T)tasks[i].ctrl[j][a];
            dt[i].yaw_ctrl[j] = (T)tasks[i].yaw_ctrl[j];
        }
        for (int a = 0; a < 3; ++a) dt[i].normal[a] = (T)tasks[i].goal_normal[a];
        dt[i].has_normal = tasks[i].has_goal_normal;
        dt[i].yaw_inc = tasks[i].yaw_increasing;
        dt[i].box_begin = tasks[i].box_begin;
        dt[i].box_count = tasks[i].box_count;
        // initial value of every optimisation variable (free waypoint rows, then the free yaw controls)
        for (int o = 0; o < 64; ++o) dt[i].xref[o] = T(0);
        for (int o = 0; o < 3 * c.n_free && o < 64; ++o) dt[i].xref[o] = (T)tasks[i].ctrl[c.first_free + o / 3][o % 3];
        for (int o = 0; o < c.n_yaw_free && 3 * c.n_free + o < 64; ++o) dt[i].xref[3 * c.n_free + o] = (T)tasks[i].yaw_ctrl[1 + o];
    }
    // Bound-pass tables: per distinct box range, the boxes sorted along a Morton curve (fp32 records,
    // pad0 = index relative to the range) and the AABBs of groups of kGroup consecutive boxes.
    {
        // Group size (even: the pair table is read group by group).  With the seed grid and the group prefilter of the split
        // kernel small groups win (round 2, neither: 16).  Measured on MI355X, clearance_kernel, groups of 4 / 6 / 8 / 16:
        // config-4 shard (2,000 boxes, S=256) 4.72 / 4.83 / 4.65 / 5.57 ms; B=2,000 S=64 cost+grad 0.487 / 0.481 / 0.453 / 0.499 ms;
        // B=300 S=101 0.263 / 0.275 / 0.269 / 0.291 ms; B=200 S=64 0.279 / 0.291 / 0.293 / 0.300 ms.
        // Ranges from 128 boxes on are grouped; from 32 on when trajectories have more than one 64-sample chunk: those launches
        // run the split kernel (launch_eval), whose prefilter makes groups pay earlier (B=64 S=256 cost+grad 0.344 -> 0.316 ms,
        // B=50 S=101 0.336 -> 0.332), while the fused one-chunk kernel loses with them (config 3 0.296 -> 0.303 ms).
        const int kGroupMin = ctx->group_min > 0 ? ctx->group_min : (c.n_safe > 64 ? 32 : 128);
        const int group_small = ctx->group_small, group_large = ctx->group_large;
        std::vector<mp::BoxRec32> sb, gr;
        std::vector<mp::BoxPair32> sp;   // the sorted boxes two per record, group by group
        std::vector<std::array<int, 8>> done;  // begin, count, sbox_begin, group_begin, group_count, n_wide, spair_begin, grid
        struct GridInfo { int begin, nx, ny, nz; float ox, oy, oz, ix, iy, iz; };
        std::vector<GridInfo> grids;     // one per box range that has a seed grid
        std::vector<int> cells;          // their cell tables, back to back
        std::unordered_map<long long, size_t> done_at;   // (begin, count) -> index in done (10,000-task batches)
        auto as_float = [](int v) { float f; std::memcpy(&f, &v, 4); return f; };
        for (int i = 0; i < ctx->n_tasks; ++i) {
            const int b0 = tasks[i].box_begin;
            int cnt = tasks[i].box_count;
            const std::array<int, 8> *hit = nullptr;
            {
                const auto it = done_at.find(((long long)b0 << 32) | (unsigned int)cnt);
                if (it != done_at.end()) hit = &done[it->second];
            }
            if (!hit) {
                // One "wide" box per range (a table, a floor: footprint at least twice the mover's circumradius each way,
                // the largest one) goes first, outside the groups: the kernel settles its z-separated pairs without an
                // exact evaluation (clearance_chunk).  Axis-aligned scenes and the box model only.
                int wide = -1;
                if (c.model == MP_MODEL_BOX && !ctx->has_rot) {
                    const double rr = 2.0 * std::sqrt(c.half_extents[0] * c.half_extents[0] + c.half_extents[1] * c.half_extents[1]);
                    double best_area = 0.0;
                    for (int k = 0; k < cnt; ++k) {
                        const double *r = &ctx->h_boxes[(size_t)(b0 + k) * 6];
                        if (rr > 0.0 && r[3] >= rr && r[4] >= rr && r[3] * r[4] > best_area) { best_area = r[3] * r[4]; wide = k; }
                    }
                }
                const int n_wide = wide >= 0 ? 1 : 0;
                std::vector<int> order;
                order.reserve((size_t)cnt);
                for (int k = 0; k < cnt; ++k)
                    if (k != wide) order.push_back(k);
                const int cnt_all = cnt;
                cnt -= n_wide;   // from here on: the boxes that go through the sorted / grouped sweep
                const int kGroup = cnt >= 1000 ? group_large : group_small;
                if (cnt >= kGroupMin) {
                    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                    for (int k : order)
                        for (int a = 0; a < 3; ++a) {
                            lo[a] = std::min(lo[a], ctx->h_boxes[(size_t)(b0 + k) * 6 + a]);
                            hi[a] = std::max(hi[a], ctx->h_boxes[(size_t)(b0 + k) * 6 + a]);
                        }
                    std::vector<unsigned long long> code((size_t)cnt_all, 0ull);
                    for (int k : order) {
                        unsigned long long c = 0;
                        unsigned q[3];
                        for (int a = 0; a < 3; ++a) {
                            double t = hi[a] > lo[a] ? (ctx->h_boxes[(size_t)(b0 + k) * 6 + a] - lo[a]) / (hi[a] - lo[a]) : 0.0;
                            q[a] = (unsigned)std::min(1023.0, std::max(0.0, t * 1024.0));
                        }
                        for (int bit = 9; bit >= 0; --bit)
                            for (int a = 0; a < 3; ++a) c = (c << 1) | ((q[a] >> bit) & 1u);
                        code[(size_t)k] = c;
                    }
                    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return code[x] < code[y]; });
                }
                // Visiting order of large scenes: whole groups of the Morton order nearest to the straight start -> goal line
                // first, so that the running minimum of a lane tightens early and later boxes are culled by the bound.
                // Any order gives the same result (minimum and lowest-index argmin are order independent).
                {
                    const double *A = tasks[i].ctrl[0], *B = tasks[i].ctrl[c.n_ctrl - 1];
                    const double ab[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
                    const double ab2 = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2];
                    auto line_gap = [&](int k) {   // centre-to-segment distance minus the half diagonal of the AABB
                        const double *r = &ctx->h_boxes[(size_t)(b0 + k) * 6];
                        double t = ab2 > 0.0 ? ((r[0] - A[0]) * ab[0] + (r[1] - A[1]) * ab[1] + (r[2] - A[2]) * ab[2]) / ab2 : 0.0;
                        t = std::min(1.0, std::max(0.0, t));
                        const double dx = r[0] - (A[0] + t * ab[0]), dy = r[1] - (A[1] + t * ab[1]), dz = r[2] - (A[2] + t * ab[2]);
                        return std::sqrt(dx * dx + dy * dy + dz * dz) - std::sqrt(r[3] * r[3] + r[4] * r[4] + r[5] * r[5]);
                    };
                    // (small scenes keep the caller's order: measured, sorting 50 boxes by this key gains nothing for a
                    // diffuse multi-start cloud, -1.5 % on config 3, and +4 % for converging CEM populations)
                    if (cnt >= kGroupMin) {
                        const int ng = (cnt + kGroup - 1) / kGroup;
                        std::vector<double> gkey((size_t)ng, 1e300);
                        for (int k = 0; k < cnt; ++k) gkey[(size_t)(k / kGroup)] = std::min(gkey[(size_t)(k / kGroup)], line_gap(order[(size_t)k]));
                        std::vector<int> gord((size_t)ng);
                        for (int g = 0; g < ng; ++g) gord[(size_t)g] = g;
                        // the last (short) group stays last so that every other group is full
                        std::stable_sort(gord.begin(), gord.end() - 1, [&](int x, int y) { return gkey[(size_t)x] < gkey[(size_t)y]; });
                        std::vector<int> reordered;
                        reordered.reserve((size_t)cnt);
                        for (int g : gord)
                            for (int k = g * kGroup; k < std::min(cnt, (g + 1) * kGroup); ++k) reordered.push_back(order[(size_t)k]);
                        order.swap(reordered);
                    }
                }
                const int sb0 = (int)sb.size(), g0 = (int)gr.size(), sp0 = (int)sp.size();
                const int gsz = cnt >= kGroupMin ? kGroup : std::max(cnt, 1);
                auto down = [](double v) { float f = (float)v; return (double)f > v ? std::nextafter(f, -INFINITY) : f; };
                auto up = [](double v) { float f = (float)v; return (double)f < v ? std::nextafter(f, INFINITY) : f; };
                auto record = [&](int orig) {
                    const double *r = &ctx->h_boxes[(size_t)(b0 + orig) * 6];
                    return mp::BoxRec32{down(r[0] - r[3]), down(r[1] - r[4]), down(r[2] - r[5]), up(r[0] + r[3]), up(r[1] + r[4]),
                                        up(r[2] + r[5]), as_float(orig), 0.f};
                };
                if (n_wide) sb.push_back(record(wide));   // record 0 of the range
                const int sbr = (int)sb.size();            // first record of the sorted rest
                std::vector<std::array<float, 3>> centre((size_t)cnt);
                for (int k = 0; k < cnt; ++k) {
                    const double *r = &ctx->h_boxes[(size_t)(b0 + order[k]) * 6];
                    sb.push_back(mp::BoxRec32{down(r[0] - r[3]), down(r[1] - r[4]), down(r[2] - r[5]), up(r[0] + r[3]), up(r[1] + r[4]),
                                              up(r[2] + r[5]), as_float(order[k]), 0.f});
                    centre[(size_t)k] = {(float)r[0], (float)r[1], (float)r[2]};
                }
                for (int k = 0; k < cnt; k += gsz) {
                    const int m = std::min(gsz, cnt - k);
                    mp::BoxRec32 g{INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY, 0.f, 0.f};
                    for (int q = 0; q < m; ++q) {
                        const mp::BoxRec32 &r = sb[(size_t)sbr + k + q];
                        g.lox = std::min(g.lox, r.lox); g.loy = std::min(g.loy, r.loy); g.loz = std::min(g.loz, r.loz);
                        g.hix = std::max(g.hix, r.hix); g.hiy = std::max(g.hiy, r.hiy); g.hiz = std::max(g.hiz, r.hiz);
                    }
                    g.pad0 = as_float(k + n_wide);   // first record of the group, relative to sbox_begin
                    g.pad1 = as_float(m);
                    gr.push_back(g);
                    // the group's boxes two by two (what the sweep reads); an odd last box shares its record with nobody
                    for (int q = 0; q < m; q += 2) {
                        const mp::BoxRec32 &r0 = sb[(size_t)sbr + k + q];
                        const bool two = q + 1 < m;
                        const mp::BoxRec32 &r1 = two ? sb[(size_t)sbr + k + q + 1] : r0;
                        mp::BoxPair32 pr;
                        pr.lox = mp::v2f{r0.lox, r1.lox}; pr.loy = mp::v2f{r0.loy, r1.loy}; pr.loz = mp::v2f{r0.loz, r1.loz};
                        pr.hix = mp::v2f{r0.hix, r1.hix}; pr.hiy = mp::v2f{r0.hiy, r1.hiy}; pr.hiz = mp::v2f{r0.hiz, r1.hiz};
                        std::memcpy(&pr.idx0, &r0.pad0, 4);
                        pr.idx1 = -1;
                        if (two) std::memcpy(&pr.idx1, &r1.pad0, 4);
                        pr.pad0 = 0.f; pr.pad1 = 0.f;
                        sp.push_back(pr);
                    }
                    // representative: the member whose centre is nearest to the centre of the group's AABB
                    const float gcx = 0.5f * (g.lox + g.hix), gcy = 0.5f * (g.loy + g.hiy), gcz = 0.5f * (g.loz + g.hiz);
                    int best = 0;
                    float bd = INFINITY;
                    for (int q = 0; q < m; ++q) {
                        const auto &c3 = centre[(size_t)k + q];
                        const float d2 = (c3[0] - gcx) * (c3[0] - gcx) + (c3[1] - gcy) * (c3[1] - gcy) + (c3[2] - gcz) * (c3[2] - gcz);
                        if (d2 < bd) { bd = d2; best = q; }
                    }
                    mp::BoxRec32 rep{};
                    if (m > 0) { rep.lox = centre[(size_t)k + best][0]; rep.loy = centre[(size_t)k + best][1]; rep.loz = centre[(size_t)k + best][2]; }
                    gr.push_back(rep);
                }
                // Seed grid of the range (clearance_chunk): a uniform grid over the AABB of the swept boxes (the wide box is
                // settled before the seeds), about grid_cells_per_box cells per box, each cell holding the box whose world
                // AABB is nearest to the cell's centre.  A sample looks up the cell it lies in and evaluates that box first:
                // its exact value bounds the sample's clearance before the sweep starts.  Only a heuristic: results do not
                // depend on what a cell holds.
                int grid_id = -1;
                // (batches of many small scenarios - 10,000 tasks of <= 13 boxes - skip it: their launches are CEM populations, which
                // bring seeds of their own, so 10,000 grid builds would only add host time to mp_set_problem)
                if (ctx->grid_mode && c.model == MP_MODEL_BOX && cnt >= 4 && (cnt >= 16 || ctx->n_tasks <= 64)) {
                    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                    for (int k : order) {
                        const double *r = &ctx->h_boxes[(size_t)(b0 + k) * 6];
                        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], r[a] - r[3 + a]); hi[a] = std::max(hi[a], r[a] + r[3 + a]); }
                    }
                    const double ext[3] = {std::max(hi[0] - lo[0], 1e-6), std::max(hi[1] - lo[1], 1e-6), std::max(hi[2] - lo[2], 1e-6)};
                    const double target = std::min(8192.0, std::max(64.0, (double)ctx->grid_cells_per_box * cnt));
                    const double h = std::cbrt(ext[0] * ext[1] * ext[2] / target);   // edge of a cubic cell
                    int nd[3];
                    for (int a = 0; a < 3; ++a) nd[a] = (int)std::min(64.0, std::max(1.0, std::ceil(ext[a] / h)));
                    GridInfo gi{(int)cells.size(), nd[0], nd[1], nd[2], (float)lo[0], (float)lo[1], (float)lo[2],
                                (float)(nd[0] / ext[0]), (float)(nd[1] / ext[1]), (float)(nd[2] / ext[2])};
                    // per box the centre and the half extents of (box + mover): a sample's clearance to a box is about the signed
                    // distance of the mover's centre to the box inflated by the mover (its footprint taken as a square of the mean of
                    // its two half sides, the yaw being unknown here); negative inside, where the deepest box is the best seed
                    const float hm[3] = {(float)(0.5 * (c.half_extents[0] + c.half_extents[1])), (float)(0.5 * (c.half_extents[0] + c.half_extents[1])),
                                         (float)c.half_extents[2]};
                    std::vector<float> bc((size_t)cnt * 3), bh((size_t)cnt * 3);
                    for (int k = 0; k < cnt; ++k) {
                        const double *r = &ctx->h_boxes[(size_t)(b0 + order[(size_t)k]) * 6];
                        for (int a = 0; a < 3; ++a) { bc[(size_t)a * cnt + k] = (float)r[a]; bh[(size_t)a * cnt + k] = (float)r[3 + a] + hm[a]; }
                    }
                    const float *cx = bc.data(), *cy = cx + cnt, *cz = cy + cnt, *ex = bh.data(), *ey = ex + cnt, *ez = ey + cnt;
                    for (int iz = 0; iz < nd[2]; ++iz)
                        for (int iy = 0; iy < nd[1]; ++iy)
                            for (int ix = 0; ix < nd[0]; ++ix) {
                                const float px = (float)(lo[0] + (ix + 0.5) * ext[0] / nd[0]), py = (float)(lo[1] + (iy + 0.5) * ext[1] / nd[1]),
                                            pz = (float)(lo[2] + (iz + 0.5) * ext[2] / nd[2]);
                                int best = 0;
                                float bd = INFINITY;
                                for (int k = 0; k < cnt; ++k) {
                                    const float qx = std::fabs(px - cx[k]) - ex[k], qy = std::fabs(py - cy[k]) - ey[k], qz = std::fabs(pz - cz[k]) - ez[k];
                                    const float ox = std::max(qx, 0.f), oy = std::max(qy, 0.f), oz = std::max(qz, 0.f);
                                    const float out2 = ox * ox + oy * oy + oz * oz;
                                    const float sd = out2 > 0.f ? std::sqrt(out2) : std::max(qx, std::max(qy, qz));
                                    if (sd < bd) { bd = sd; best = k; }
                                }
                                cells.push_back(order[(size_t)best]);
                            }
                    grid_id = (int)grids.size();
                    grids.push_back(gi);
                }
                done.push_back({b0, cnt_all, sb0, g0 / 2, ((int)gr.size() - g0) / 2, n_wide, sp0, grid_id});
                done_at[((long long)b0 << 32) | (unsigned int)cnt_all] = done.size() - 1;
                hit = &done.back();
            }
            dt[i].sbox_begin = (*hit)[2];
            dt[i].group_begin = (*hit)[3];
            dt[i].group_count = (*hit)[4];
            dt[i].n_wide = (*hit)[5];
            dt[i].spair_begin = (*hit)[6];
            dt[i].grid_begin = -1;
            dt[i].gnx = dt[i].gny = dt[i].gnz = 1;
            dt[i].gox = dt[i].goy = dt[i].goz = 0.f;
            dt[i].gix = dt[i].giy = dt[i].giz = 0.f;
            if ((*hit)[7] >= 0) {
                const GridInfo &gi = grids[(size_t)(*hit)[7]];
                dt[i].grid_begin = gi.begin; dt[i].gnx = gi.nx; dt[i].gny = gi.ny; dt[i].gnz = gi.nz;
                dt[i].gox = gi.ox; dt[i].goy = gi.oy; dt[i].goz = gi.oz; dt[i].gix = gi.ix; dt[i].giy = gi.iy; dt[i].giz = gi.iz;
            }
        }
        if (cells.empty()) cells.push_back(-1);
        if ((rc = upload(ctx, ctx->grid, cells.data(), cells.size() * sizeof(int)))) return rc;
        if ((rc = upload(ctx, ctx->boxes32, sb.data(), sb.size() * sizeof(mp::BoxRec32)))) return rc;
        if ((rc = upload(ctx, ctx->groups, gr.data(), gr.size() * sizeof(mp::BoxRec32)))) return rc;
        if ((rc = upload(ctx, ctx->pairs32, sp.data(), sp.size() * sizeof(mp::BoxPair32)))) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if ((rc = upload(ctx, ctx->d_tasks, dt.data(), dt.size() * sizeof(mp::DevTask<T>)))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // locals go out of scope
    return MP_OK;
}

template <typename T, int MODEL, bool GRAD, int ROT, bool PRE>
int launch_eval_p(mp_ctx *ctx, const mp::EvalArgs<T> &args, int wpb, int blocks, size_t lds_bytes)
{
    void (*kern)(const mp::EvalArgs<T>) = mp::eval_kernel<T, MODEL, GRAD, ROT, PRE>;
    if constexpr (sizeof(T) == 8 && MODEL == 1 && GRAD && ROT == 0 && !PRE) {
        if (args.five_waves) kern = mp::eval_kernel<T, MODEL, GRAD, ROT, PRE, 5>;
    }
    if constexpr (sizeof(T) == 8 && MODEL == 1 && !GRAD && ROT == 0 && !PRE) {
        if (args.five_waves) kern = mp::eval_kernel<T, MODEL, GRAD, ROT, PRE, 6>;
    }
    if constexpr (MODEL == 1 && ROT < 2 && PRE && GRAD) {
        if (args.wide == 4) kern = mp::eval_kernel<T, MODEL, GRAD, ROT, PRE, 0, 4>;
        if (args.wide == 3) kern = mp::eval_kernel<T, MODEL, GRAD, ROT, PRE, 0, 3>;
    }
    if (lds_bytes > 65536) {
        // the opt-in for > 64 KB of dynamic LDS is per kernel variant and device: remembered per (device, kernel), raised when a
        // larger slice comes.  Contexts on several host threads share the table (one mutex; the path is taken by launches
        // of more than ~540 fp64 samples only).
        static std::mutex grant_mutex;
        static std::unordered_map<const void *, int> granted[64];
        const void *key = reinterpret_cast<const void *>(kern);
        std::lock_guard<std::mutex> lock(grant_mutex);
        int &g = granted[ctx->device & 63][key];
        if ((int)lds_bytes > g) {
            HIP_TRY(ctx, hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            g = (int)lds_bytes;
        }
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * wpb), lds_bytes, ctx->stream, args);
    HIP_TRY(ctx, hipGetLastError());
    return MP_OK;
}

template <typename T, int MODEL, bool GRAD, int ROT>
int launch_eval_r(mp_ctx *ctx, const mp::EvalArgs<T> &args, int wpb, int blocks, size_t lds_bytes)
{
    if (MODEL == 1 && args.pre_d) return launch_eval_p<T, MODEL, GRAD, ROT, MODEL == 1>(ctx, args, wpb, blocks, lds_bytes);
    return launch_eval_p<T, MODEL, GRAD, ROT, false>(ctx, args, wpb, blocks, lds_bytes);
}

template <typename T, int MODEL, bool GRAD>
int launch_eval_k(mp_ctx *ctx, const mp::EvalArgs<T> &args, int wpb, int blocks, size_t lds_bytes)
{
    if (ctx->has_rot == 2) return launch_eval_r<T, MODEL, GRAD, 2>(ctx, args, wpb, blocks, lds_bytes);
    return ctx->has_rot ? launch_eval_r<T, MODEL, GRAD, 1>(ctx, args, wpb, blocks, lds_bytes)
                        : launch_eval_r<T, MODEL, GRAD, 0>(ctx, args, wpb, blocks, lds_bytes);
}

template <typename T>
int launch_eval(mp_ctx *ctx, long long n, const void *X, const int32_t *task_of, void *cost, void *terms, void *grad,
                void *minc, uint8_t *flags, void *dist, int group_size = 0, const int *group_task = nullptr,
                const uint8_t *group_done = nullptr, void *ext_scratch = nullptr, const int *seeds = nullptr, bool seed_pass = false)
{
    const mp_config &c = ctx->cfg;
    mp::EvalArgs<T> a{};
    a.S = c.n_samples; a.C = c.n_ctrl; a.ff = c.first_free; a.nf = c.n_free; a.Cy = c.n_yaw_ctrl;
    a.nyf = c.n_yaw_free; a.Sp = c.n_safe; a.last_cond = c.last_if_not_close; a.keep_n = c.keep_n;
    a.use_pref = c.use_pref; a.tail_start = c.tail_start; a.use_goal_target = c.use_goal_target;
    a.goal_tail_n = c.goal_tail_n; a.mask = c.term_mask;
    a.w_elems = mp::w_elems_for(c.n_ctrl, c.n_yaw_ctrl);
    a.lds_per_wave = mp::lds_bytes_per_wave(c.n_samples, (int)sizeof(T), grad != nullptr, a.w_elems);
    a.n_tasks = ctx->n_tasks;
    for (int i = 0; i < MP_NTERMS; ++i) {
        a.w[i] = (T)c.weights[i];
        a.w2[i] = a.w[i] * T(2);
        a.wf[i] = (c.term_mask & (1u << i)) ? a.w[i] : T(0);
    }
    a.req = (T)c.required_clearance; a.pref = (T)c.preferred_clearance; a.appr_tgt = (T)c.approach_target;
    a.goal_tgt = (T)c.goal_clearance_target; a.radius = (T)c.point_radius;
    a.ax = (T)c.half_extents[0]; a.ay = (T)c.half_extents[1]; a.az = (T)c.half_extents[2];
    a.trig = mp::trig_table<T>();
    a.du = T(1) / (T)(c.n_samples - 1); a.inv_du = T(1) / a.du; a.inv_h2 = T(1) / (T(2) * a.du);
    a.wf[1] = c.n_samples >= 3 ? a.wf[1] * a.du : T(0);   // curvature: the sum is weighted with w_curv du
    if (c.n_samples < 3) a.wf[2] = T(0);
    a.idx_safe = (const int *)ctx->d_idx.p; a.inv_idx = (const int *)ctx->d_inv.p; a.idx_identity = ctx->idx_identity; a.amask = (const uint8_t *)ctx->d_mask.p;
    a.M = (const T *)ctx->d_M.p; a.My = (const T *)ctx->d_My.p; a.tsched = (const T *)ctx->d_tsched.p;
    a.tasks = (const mp::DevTask<T> *)ctx->d_tasks.p; a.boxes = (const mp::BoxRec<T> *)ctx->boxes.p;
    a.boxes32 = (const mp::BoxRec32 *)ctx->boxes32.p; a.pairs = (const mp::BoxPair32 *)ctx->pairs32.p; a.groups = (const mp::BoxRec32 *)ctx->groups.p;
    a.scene_radius = ctx->scene_radius;
    a.grid = ctx->grid_mode ? (const int *)ctx->grid.p : nullptr;
    a.tilt = (const T *)ctx->tilt.p;
    a.X = (const T *)X; a.task_of = task_of; a.cpt = ctx->cands_per_task; a.n = n;
    a.group_size = group_size; a.group_task = group_task; a.group_done = group_done;
    a.cost = (T *)cost; a.terms = (T *)terms; a.grad = (T *)grad; a.minc = (T *)minc; a.dist = (T *)dist;
    a.flags = flags;
    a.seed = (group_size > 0 && grad == nullptr && !seed_pass) ? seeds : nullptr;
    // exact passes shared between the chunks of a trajectory (clearance_chunk): a queue entry then holds the trajectory's sample
    // index; 16-bit entries (compact instances) have 7 + 9 bits for it and the box, 32-bit entries 10 + 22
    a.defer = 0;   // 1: entries of either width have the room, 2: 32-bit entries only
    if (ctx->defer_mode && c.n_safe > 64 && c.n_safe <= 1024 && ctx->max_task_boxes < (1 << 22))
        a.defer = (c.n_safe <= 128 && ctx->max_task_boxes < 512) ? 1 : 2;
    if (seed_pass) {
        // the mean trajectory of every CEM instance through clearance_kernel: its table of deciding boxes is the seed table of
        // the instance's next population (one row per instance: group_size == 1 here)
        a.pre_d = (T *)ctx->d_seed_d.p;
        a.pre_arg = (int *)ctx->d_seed.p;
        const long long units = n * ((c.n_safe + 63) / 64);
        const int cblocks = (int)std::min<long long>((units + 3) / 4, (long long)ctx->num_cu * 128);
        const size_t clds = (size_t)4 * mp::clearance_lds_bytes_per_wave((int)sizeof(T));
        if (ctx->has_rot == 2) hipLaunchKernelGGL((mp::clearance_kernel<T, 2>), dim3(cblocks), dim3(256), clds, ctx->stream, a);
        else if (ctx->has_rot) hipLaunchKernelGGL((mp::clearance_kernel<T, 1>), dim3(cblocks), dim3(256), clds, ctx->stream, a);
        else hipLaunchKernelGGL((mp::clearance_kernel<T, 0>), dim3(cblocks), dim3(256), clds, ctx->stream, a);
        HIP_TRY(ctx, hipGetLastError());
        return MP_OK;
    }

    // Five-wave instance of the fp64 gradient kernel (mp_kernels.hpp, WV = 5): fused axis-aligned box scenes with one
    // 64-sample chunk, < 1024 boxes per task and a waypoint table of <= 256 bytes, when five four-wave blocks of its compact
    // slice fit the CU's LDS (handed out in granules of 1,280 bytes, 128 per CU)
    {
        const bool will_split = c.model == MP_MODEL_BOX &&
                                (ext_scratch || ctx->split_mode == 1 ||
                                 (ctx->split_mode < 0 && (ctx->max_task_boxes >= 128 || (c.n_safe > 64 && ctx->max_task_boxes >= 32))));
        const size_t granules = ctx->max_lds / 1280;
        auto blocks_fit = [&](size_t block_bytes) { return granules / ((block_bytes + 1279) / 1280); };
        const bool compact_ok = sizeof(T) == 8 && terms == nullptr && c.model == MP_MODEL_BOX && !ctx->has_rot && !will_split &&
                                ctx->max_task_boxes < 1024 && a.w_elems * (int)sizeof(T) <= 256 && ctx->five_wave_mode != 0;
        const size_t five_block = (size_t)4 * mp::lds_bytes_per_wave_five(c.n_samples);
        a.five_waves = (compact_ok && grad != nullptr && c.n_samples <= 64 && blocks_fit(five_block) >= 5) ? 1 : 0;
        if (a.five_waves) a.lds_per_wave = mp::lds_bytes_per_wave_five(c.n_samples);
        // ... and of the fp64 cost-only kernel (WV = 6: six waves per SIMD, no term breakdown, no deciding boxes, any number
        // of chunks): taken when its smaller slice puts more four-wave blocks on a CU than the standard one (five)
        if (compact_ok && grad == nullptr) {
            const size_t cblock = (size_t)4 * mp::lds_bytes_per_wave_compact_cost(c.n_samples);
            const size_t std_blocks = std::min<size_t>(blocks_fit((size_t)4 * a.lds_per_wave), 5);
            if (std::min<size_t>(blocks_fit(cblock), 6) > std_blocks) {
                a.five_waves = 1;
                a.lds_per_wave = mp::lds_bytes_per_wave_compact_cost(c.n_samples);
            }
        }
    }
    const size_t per_wave = (size_t)a.lds_per_wave;
    if (per_wave > ctx->max_lds)
        return fail(ctx, MP_E_UNSUPPORTED, "n_samples_curve=%d needs %zu B of LDS per wave (limit %zu)", c.n_samples,
                    per_wave, ctx->max_lds);
    // waves per block: a block's waves are spread over the CU's four SIMDs, several small blocks need not be, so the
    // largest block (up to four waves) wins unless a smaller one puts clearly more waves on the CU (blocks share its LDS).
    // Measured: S=101 with gradient, 4 blocks of 2 waves 0.87 ms, 2 blocks of 4 waves 0.78 ms; point model S=121 cost only,
    // 6 blocks of 3 waves (18 waves) 0.176 ms, 4 blocks of 4 (16 waves) 0.156 ms.
    int wpb = 1;
    if (a.five_waves) wpb = 4;
    else {
        size_t waves_of[5] = {0, 0, 0, 0, 0}, best = 0;
        // VGPR budget of the kernel variant (__launch_bounds__): waves per SIMD x 4 SIMDs per CU
        const size_t vgpr_cap = (size_t)4 * (grad != nullptr ? mp::eval_min_waves<T, true>() : mp::eval_min_waves<T, false>());
        for (int w = 1; w <= 4; ++w) {
            if (per_wave * w > ctx->max_lds) break;
            // blocks that fit: LDS is handed out in granules of 1,280 bytes (a 54,144-byte block takes 43 of the CU's 128)
            const size_t fit = (ctx->max_lds / 1280) / ((per_wave * w + 1279) / 1280);
            waves_of[w] = std::min<size_t>(std::min<size_t>(fit, 16) * w, vgpr_cap);
            best = std::max(best, waves_of[w]);
        }
        for (int w = 1; w <= 4; ++w)
            if (waves_of[w] > 0 && waves_of[w] * 100 >= best * MP_WPB_PCT) wpb = w;
        // At low occupancy (<= 12 resident waves) two-wave blocks are taken when they put a fifth more waves on the CU than
        // four-wave blocks and more than three-wave blocks (S = 108 ... 121 with gradient: 10 waves against 8 or 9, +7 ... +10 %).
        if (MP_WPB_TWO && wpb >= 3 && best <= 12 && waves_of[4] > 0 && waves_of[2] * 5 >= waves_of[4] * 6 && waves_of[2] > waves_of[3]) wpb = 2;
    }
    const size_t lds_bytes = per_wave * wpb;
    long long need = (n + wpb - 1) / wpb;
#ifndef MP_WIDE_BLOCKS_PER_CU
#define MP_WIDE_BLOCKS_PER_CU 128   // config-4 shard, wide eval_kernel: 16 / 128 / 512 -> 1.56 / 1.46 / 1.48 ms
#endif
#ifndef MP_CLR_WPB
#define MP_CLR_WPB 4   // waves per block of clearance_kernel (config-4 shard: 1 / 2 / 4 -> 8.65 / 8.30 / 8.34 ms)
#endif
#ifndef MP_CLR_BLOCKS_PER_CU
#define MP_CLR_BLOCKS_PER_CU 128    // config-4 shard, clearance_kernel: 32 / 64 / 128 / 256 / 512 -> 8.56 / 8.39 / 8.32 / 8.42 / 8.58 ms
#endif
#ifndef MP_BLOCKS_PER_CU
#define MP_BLOCKS_PER_CU 128   // grid cap of eval_kernel per CU (grid-stride beyond it).  Measured at config 3, 65,536 candidates:
                               // 5 / 10 / 16 / 32 / 64 / 128 / 1024 per CU -> 0.397 / 0.373 / 0.361 / 0.352 / 0.352 / 0.350 / 0.351 ms;
                               // 1,048,576 candidates: 16 / 32 / 64 / 128 / 1024 -> 5.65 / 5.43 / 5.33 / 5.27 / 5.40 ms
#endif
    int blocks = (int)std::min<long long>(need, (long long)ctx->num_cu * MP_BLOCKS_PER_CU);
    if (blocks < 1) return MP_OK;
    // Split mode: the clearance profile is computed by clearance_kernel first (full occupancy whatever S is, bound
    // seeded from group representatives) when the scene is large enough to have box groups (>= 128 boxes in a
    // task), or when a trajectory has more than one 64-sample chunk and the scene is not tiny.  Measured on
    // MI355X (tools/shape_bench.py): B=200 S=64 -9 %, B=2000 S=64 -13 %, B=50 S=256 -32 %, B=2000 S=256 -75 %;
    // B=13 S=101 and B=50 S=64 are faster fused.
    const int nw = std::min(4, std::max(3, (c.n_samples + 63) / 64));
    const int D_out = 3 * c.n_free + c.n_yaw_free;
    const size_t wide_bytes = (size_t)mp::wide_block_bytes(c.n_samples, (int)sizeof(T), nw, D_out, a.w_elems);
    const bool wide_ok = c.model == MP_MODEL_BOX && grad != nullptr && terms == nullptr && ctx->has_rot < 2 &&
                         c.n_samples > ctx->wide_min_s && ctx->wide_mode != 0 && wide_bytes <= ctx->max_lds;
    if (c.model == MP_MODEL_BOX) {
        const size_t pre_bytes = (size_t)n * (size_t)c.n_safe * (sizeof(T) + sizeof(int));
        bool split = ctx->split_mode == 1 || (ctx->split_mode < 0 && (ctx->max_task_boxes >= 128 || (c.n_safe > 64 && ctx->max_task_boxes >= 32)));
        // ... and whenever the wide instance can take the launch: on long trajectories split + wide beats the fused kernel even
        // for a handful of boxes (B=13: S=192 0.367 -> 0.276 ms, S=256 0.565 -> 0.321, S=512 0.998 -> 0.362; B=5 S=256 0.469 -> 0.289)
        if (ctx->split_mode < 0 && wide_ok && ctx->max_task_boxes >= 1) split = true;
        if (ext_scratch) split = true;   // mp_eval_noalloc: always the split kernels, tables in the caller-independent static array
        if (split && pre_bytes <= ((size_t)16 << 30)) {
            if (ext_scratch) {
                a.pre_d = (T *)ext_scratch;
                a.pre_arg = (int *)((unsigned char *)ext_scratch + noalloc_arg_offset((size_t)n * c.n_safe * sizeof(T)));
            } else {
                int rc = reserve(ctx, ctx->s_pre_d, (size_t)n * c.n_safe * sizeof(T));
                if (rc) return rc;
                if ((rc = reserve(ctx, ctx->s_pre_arg, (size_t)n * c.n_safe * sizeof(int)))) return rc;
                a.pre_d = (T *)ctx->s_pre_d.p;
                a.pre_arg = (int *)ctx->s_pre_arg.p;
            }
            const long long units = n * ((c.n_safe + 63) / 64);
            const int cblocks = (int)std::min<long long>((units + MP_CLR_WPB - 1) / MP_CLR_WPB, (long long)ctx->num_cu * MP_CLR_BLOCKS_PER_CU);
            const size_t clds = (size_t)MP_CLR_WPB * mp::clearance_lds_bytes_per_wave((int)sizeof(T));
            if (ctx->has_rot == 2)
                hipLaunchKernelGGL((mp::clearance_kernel<T, 2>), dim3(cblocks), dim3(64 * MP_CLR_WPB), clds, ctx->stream, a);
            else if (ctx->has_rot)
                hipLaunchKernelGGL((mp::clearance_kernel<T, 1>), dim3(cblocks), dim3(64 * MP_CLR_WPB), clds, ctx->stream, a);
            else if (grad != nullptr)
                hipLaunchKernelGGL((mp::clearance_kernel<T, 0>), dim3(cblocks), dim3(64 * MP_CLR_WPB), clds, ctx->stream, a);
            else   // cost only (a CEM population on a large scene): no deciding boxes
                hipLaunchKernelGGL((mp::clearance_kernel<T, 0, false>), dim3(cblocks), dim3(64 * MP_CLR_WPB), clds, ctx->stream, a);
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    // Wide instance (mp_kernels.hpp, NW = 3 / 4) for split-mode gradient launches without the term breakdown on long
    // trajectories: there the slice of a candidate leaves room for one or two waves per SIMD; one candidate per three- or
    // four-wave block keeps that many waves per slice busy.  Measured (B=50, cost+grad, standard -> wide): S=129 0.548 -> 0.583 ms,
    // S=160 0.586 -> 0.611, S=192 0.360 -> 0.320 (four waves: 0.340), S=256 0.587 -> 0.433, S=512 0.788 -> 0.437; two-wave
    // blocks at S=101 lose (0.371 -> 0.410), and cost-only launches lose 1-5 % (their slices are half the size).  fp32 (slices
    // half the size, 7-wave register budget): S=192 +10 %, S=256 +16 %, S=512 +54 %.
    size_t lds_launch = lds_bytes;
    if (wide_ok && a.pre_d) {
        // its block: records | deciding boxes | RED, no waypoint table in front and no work queue (mp_kernels.hpp, wide_block_bytes):
        // 32,000 B at S = 256 with up to 8 outputs, so that five blocks fit the CU's 128 LDS granules of 1,280 B
        a.wide = nw;
        a.wide_rs = mp::wide_red_stride(D_out);
        a.lds_per_wave = mp::wide_red_offset(c.n_samples, (int)sizeof(T));
        wpb = nw;
        lds_launch = wide_bytes;
        blocks = (int)std::min<long long>(n, (long long)ctx->num_cu * MP_WIDE_BLOCKS_PER_CU);
    }
    const bool g = grad != nullptr;
    if (c.model == MP_MODEL_BOX)
        return g ? launch_eval_k<T, 1, true>(ctx, a, wpb, blocks, lds_launch)
                 : launch_eval_k<T, 1, false>(ctx, a, wpb, blocks, lds_launch);
    return g ? launch_eval_k<T, 0, true>(ctx, a, wpb, blocks, lds_bytes)
             : launch_eval_k<T, 0, false>(ctx, a, wpb, blocks, lds_bytes);
}

template <typename T>
int cem_solve(mp_ctx *ctx, long long n_inst, const int32_t *d_task, const T *d_x0, const T *d_s0, const mp_cem_options &o,
              const T *d_noise, T *best_x, T *best_val, int32_t *nit, T *best_minc, uint8_t *best_flags, int mem)
{
    const mp_config &c = ctx->cfg;
    const int D = 3 * c.n_free + c.n_yaw_free, pop = o.population_size;
    const int n_elite = std::max(2, (int)std::ceil(pop * o.elite_frac));
    const size_t nd = (size_t)n_inst * D, np_ = (size_t)n_inst * pop;
    int rc;
    if ((rc = reserve(ctx, ctx->c_X, np_ * D * sizeof(T)))) return rc;
    if ((rc = reserve(ctx, ctx->c_vals, np_ * sizeof(T)))) return rc;
    if ((rc = reserve(ctx, ctx->c_minc, np_ * sizeof(T)))) return rc;
    if ((rc = reserve(ctx, ctx->c_flags, np_))) return rc;
    if ((rc = reserve(ctx, ctx->c_mu, nd * sizeof(T)))) return rc;
    if ((rc = reserve(ctx, ctx->c_sigma, nd * sizeof(T)))) return rc;
    if ((rc = reserve(ctx, ctx->c_bestx, nd * sizeof(T)))) return rc;
    if ((rc = reserve(ctx, ctx->c_bestval, (size_t)n_inst * sizeof(T)))) return rc;
    if ((rc = reserve(ctx, ctx->c_prev, (size_t)n_inst * sizeof(T)))) return rc;
    if ((rc = reserve(ctx, ctx->c_bminc, (size_t)n_inst * sizeof(T)))) return rc;
    if ((rc = reserve(ctx, ctx->c_bflags, (size_t)n_inst))) return rc;
    if ((rc = reserve(ctx, ctx->c_done, (size_t)n_inst))) return rc;
    if ((rc = reserve(ctx, ctx->c_nit, (size_t)n_inst * sizeof(int)))) return rc;
    if ((rc = reserve(ctx, ctx->c_active, 256))) return rc;
    // seeds of the clearance bound (mp_kernels.hpp, clearance_chunk): the deciding boxes of every instance's mean trajectory,
    // refreshed after every update of the mean (1 / population of an evaluation's work)
    const bool seeded = ctx->seed_mode && c.model == MP_MODEL_BOX && ctx->max_task_boxes >= 2;
    if (seeded) {
        if ((rc = reserve(ctx, ctx->d_seed, (size_t)n_inst * c.n_safe * sizeof(int)))) return rc;
        if ((rc = reserve(ctx, ctx->d_seed_d, (size_t)n_inst * c.n_safe * sizeof(T)))) return rc;
    }
    auto seed_pass = [&]() -> int {
        if (!seeded) return MP_OK;
        return launch_eval<T>(ctx, n_inst, ctx->c_mu.p, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, d_task,
                              (const uint8_t *)ctx->c_done.p, nullptr, nullptr, true);
    };
    HIP_TRY(ctx, hipMemsetAsync(ctx->c_active.p, 0, 256, ctx->stream));   // [0] running count, [1] generation counter
    mp::CemState<T> st{(T *)ctx->c_mu.p, (T *)ctx->c_sigma.p, (T *)ctx->c_bestx.p, (T *)ctx->c_bestval.p,
                       (T *)ctx->c_prev.p, (T *)ctx->c_bminc.p, (uint8_t *)ctx->c_bflags.p, (uint8_t *)ctx->c_done.p,
                       (int *)ctx->c_nit.p, (int *)ctx->c_active.p, (int *)ctx->c_active.p + 1};
    const int eb = (int)std::min<long long>(((long long)nd + 255) / 256, (long long)ctx->num_cu * 8);
    hipLaunchKernelGGL(mp::cem_init_kernel<T>, dim3(std::max(eb, 1)), dim3(256), 0, ctx->stream, n_inst, D, st, d_x0, d_s0,
                       (T)o.min_sigma);
    HIP_TRY(ctx, hipGetLastError());
    if ((rc = seed_pass())) return rc;   // first population: seeds from x0
    const int sb = (int)std::min<long long>(((long long)np_ * D + 255) / 256, (long long)ctx->num_cu * 16);
    const int wpb = 4;
    const int ub = (int)std::min<long long>((n_inst + wpb - 1) / wpb, (long long)ctx->num_cu * 16);
    const size_t ulds = (size_t)wpb * pop * sizeof(T);
    // one generation: sample, evaluate, tick (generation counter + running count), update — all on the stream
    auto generation = [&](const T *noise_g) -> int {
        hipLaunchKernelGGL(mp::cem_sample_kernel<T>, dim3(sb), dim3(256), 0, ctx->stream, n_inst, pop, D, st, noise_g,
                           (unsigned long long)o.seed, (unsigned long long)o.instance_base, (T *)ctx->c_X.p);
        if (hipGetLastError() != hipSuccess) return MP_E_HIP;
        int r = launch_eval<T>(ctx, (long long)np_, ctx->c_X.p, nullptr, ctx->c_vals.p, nullptr, nullptr, ctx->c_minc.p,
                               (uint8_t *)ctx->c_flags.p, nullptr, pop, d_task, (const uint8_t *)ctx->c_done.p, nullptr,
                               seeded ? (const int *)ctx->d_seed.p : nullptr);
        if (r) return r;
        hipLaunchKernelGGL(mp::cem_tick_kernel<T>, dim3(1), dim3(64), 0, ctx->stream, st);
        hipLaunchKernelGGL(mp::cem_update_kernel<T>, dim3(ub), dim3(64 * wpb), ulds, ctx->stream, n_inst, pop, D, n_elite,
                           (T)o.alpha, (T)o.min_sigma, (T)o.tol, st, (const T *)ctx->c_X.p, (const T *)ctx->c_vals.p,
                           (const T *)ctx->c_minc.p, (const uint8_t *)ctx->c_flags.p);
        if (hipGetLastError() != hipSuccess) return MP_E_HIP;
        return seed_pass();
    };
    auto still_running = [&](int &active) -> int {
        HIP_TRY(ctx, hipMemcpyAsync(&active, ctx->c_active.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return MP_OK;
    };
    int gen = 0;
    if (o.max_iter > 0) {   // the first generation runs plainly: it performs every allocation the launches need
        if ((rc = generation(d_noise))) return rc == MP_E_HIP ? fail(ctx, rc, "mp_cem_solve: kernel launch failed") : rc;
        gen = 1;
    }
    // The launch-bound part of a small solve is the generation loop (four launches per generation): capture four
    // generations into a HIP graph once and replay it; the host polls the running count after every replay.  Any
    // failure to capture (e.g. the context is bound to the legacy default stream) falls back to plain launches;
    // the results are the same either way.
    constexpr int kGraphGens = 4;
    hipGraphExec_t exec = nullptr;
    if (!d_noise && ctx->cem_graph && o.max_iter >= 1 + 2 * kGraphGens) {
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            int crc = MP_OK;
            for (int k = 0; k < kGraphGens && crc == MP_OK; ++k) crc = generation(nullptr);
            hipGraph_t graph = nullptr;
            const hipError_t ee = hipStreamEndCapture(ctx->stream, &graph);
            if (crc == MP_OK && ee == hipSuccess && graph) {
                if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
            }
            if (graph) hipGraphDestroy(graph);
        }
        (void)hipGetLastError();   // a refused capture leaves a sticky error behind
    }
    bool stop = false;
    while (gen < o.max_iter && !stop) {
        if (exec && gen + kGraphGens <= o.max_iter) {
            if (hipGraphLaunch(exec, ctx->stream) != hipSuccess) { hipGraphExecDestroy(exec); exec = nullptr; (void)hipGetLastError(); continue; }
            gen += kGraphGens;
        } else {
            const T *noise_g = d_noise ? d_noise + (size_t)gen * np_ * D : nullptr;
            if ((rc = generation(noise_g))) {
                if (exec) hipGraphExecDestroy(exec);
                return rc == MP_E_HIP ? fail(ctx, rc, "mp_cem_solve: kernel launch failed") : rc;
            }
            gen += 1;
        }
        if (exec || (gen & 3) == 0 || gen == o.max_iter) {   // poll the running count
            int active = 0;
            if ((rc = still_running(active))) { if (exec) hipGraphExecDestroy(exec); return rc; }
            stop = active == 0;
        }
    }
    if (exec) hipGraphExecDestroy(exec);
    const hipMemcpyKind kind = mem == MP_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    HIP_TRY(ctx, hipMemcpyAsync(best_x, ctx->c_bestx.p, nd * sizeof(T), kind, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(best_val, ctx->c_bestval.p, (size_t)n_inst * sizeof(T), kind, ctx->stream));
    if (nit) HIP_TRY(ctx, hipMemcpyAsync(nit, ctx->c_nit.p, (size_t)n_inst * sizeof(int), kind, ctx->stream));
    if (best_minc) HIP_TRY(ctx, hipMemcpyAsync(best_minc, ctx->c_bminc.p, (size_t)n_inst * sizeof(T), kind, ctx->stream));
    if (best_flags) HIP_TRY(ctx, hipMemcpyAsync(best_flags, ctx->c_bflags.p, (size_t)n_inst, kind, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MP_OK;
}

// Every entry point runs with the context's GPU current and puts the caller's device back on return (a context on
// cuda:1 must not move a thread whose torch device is cuda:0).
struct DeviceGuard {
    int prev = -1;
    int enter(mp_ctx *ctx)
    {
        if (!ctx) return MP_E_ARG;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); cur = -1; }
        if (cur == ctx->device) return MP_OK;
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess) return fail(ctx, MP_E_HIP, "hipSetDevice(%d): %s", ctx->device, hipGetErrorString(e));
        prev = cur;
        return MP_OK;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

extern "C" {

int mp_api_version(void) { return MP_API_VERSION; }

const char *mp_last_error(const mp_ctx *ctx) { return ctx ? ctx->err : g_create_err; }

int mp_create(int device_id, int dtype, mp_ctx **out)
{
    if (!out) return fail(nullptr, MP_E_ARG, "mp_create: out is NULL");
    *out = nullptr;
    if (dtype != MP_F64 && dtype != MP_F32) return fail(nullptr, MP_E_ARG, "mp_create: dtype must be MP_F64 or MP_F32");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, MP_E_HIP, "mp_create: no HIP device available (%s); this library has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device_id < 0 || device_id >= count)
        return fail(nullptr, MP_E_ARG, "mp_create: device_id %d out of range (0..%d)", device_id, count - 1);
    struct Restore {   // mp_create leaves the caller's current device as it found it
        int prev = -1;
        Restore() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
        ~Restore() { if (prev >= 0) (void)hipSetDevice(prev); }
    } restore_;
    mp_ctx *ctx = new mp_ctx();
    ctx->device = device_id;
    if (const char *sm = std::getenv("MP_SPLIT_CLEARANCE")) ctx->split_mode = std::atoi(sm) != 0 ? 1 : 0;
    if (const char *sm = std::getenv("MP_SEEDS")) ctx->seed_mode = std::atoi(sm) != 0 ? 1 : 0;
    if (const char *gm = std::getenv("MP_GRID")) ctx->grid_mode = std::atoi(gm) != 0 ? 1 : 0;
    if (const char *gm = std::getenv("MP_GROUP_MIN")) ctx->group_min = std::max(4, std::atoi(gm));
    if (const char *gs = std::getenv("MP_GROUP_SMALL")) ctx->group_small = std::max(2, std::atoi(gs) & ~1);
    if (const char *gl = std::getenv("MP_GROUP_LARGE")) ctx->group_large = std::max(2, std::atoi(gl) & ~1);
    if (const char *gc = std::getenv("MP_GRID_CELLS")) ctx->grid_cells_per_box = std::max(1, std::atoi(gc));
    if (const char *sm = std::getenv("MP_DEFER")) ctx->defer_mode = std::atoi(sm) != 0 ? 1 : 0;
    if (const char *wm = std::getenv("MP_WIDE")) ctx->wide_mode = std::atoi(wm) != 0 ? -1 : 0;
    if (const char *ws = std::getenv("MP_WIDE_MIN_S")) ctx->wide_min_s = std::atoi(ws);
    if (const char *fw = std::getenv("MP_FIVE_WAVES")) ctx->five_wave_mode = std::atoi(fw) != 0 ? -1 : 0;
    if (const char *cg = std::getenv("MP_CEM_GRAPH")) ctx->cem_graph = std::atoi(cg) != 0 ? 1 : 0;
    ctx->dtype = dtype;
    auto bail = [&](const char *what, hipError_t err) {
        fail(nullptr, MP_E_HIP, "mp_create: %s: %s", what, hipGetErrorString(err));
        delete ctx;
        return MP_E_HIP;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->max_lds = prop.maxSharedMemoryPerMultiProcessor > 0 ? (size_t)prop.maxSharedMemoryPerMultiProcessor : 65536;
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    ctx->stream = ctx->own_stream;
    if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    if (device_id < 64) {
        // this GPU's copy of the static scratch array of mp_eval_noalloc (device_id is current here); contexts may be created
        // on several host threads at once
        std::lock_guard<std::mutex> lock(g_scratch_order[device_id].m);
        if (!g_scratch_ptr[device_id].load(std::memory_order_acquire)) {
            void *ptr = nullptr;
            if ((e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch))) != hipSuccess) return bail("hipGetSymbolAddress(scratch)", e);
            if ((e = hipEventCreateWithFlags(&g_scratch_order[device_id].done, hipEventDisableTiming)) != hipSuccess)
                return bail("hipEventCreate(scratch order)", e);
            g_scratch_ptr[device_id].store(ptr, std::memory_order_release);
        }
    }
    *out = ctx;
    return MP_OK;
}

int mp_destroy(mp_ctx *ctx)
{
    if (!ctx) return MP_OK;
    DeviceGuard guard_;
    (void)guard_.enter(ctx);
    hipStreamSynchronize(ctx->stream);
    DevBuf *bufs[] = {&ctx->boxes, &ctx->boxes32, &ctx->pairs32, &ctx->groups, &ctx->grid, &ctx->d_idx, &ctx->d_inv, &ctx->d_mask, &ctx->d_M, &ctx->d_My, &ctx->d_tsched, &ctx->d_tasks,
                      &ctx->s_X, &ctx->s_taskof, &ctx->s_cost, &ctx->s_terms, &ctx->s_grad, &ctx->s_minc,
                      &ctx->s_flags, &ctx->s_dist, &ctx->s_aux0, &ctx->s_aux1, &ctx->s_pre_d, &ctx->s_pre_arg, &ctx->d_seed, &ctx->d_seed_d, &ctx->tilt, &ctx->c_X, &ctx->c_vals, &ctx->c_minc,
                      &ctx->c_flags, &ctx->c_mu, &ctx->c_sigma, &ctx->c_bestx, &ctx->c_bestval, &ctx->c_prev, &ctx->c_bminc,
                      &ctx->c_bflags, &ctx->c_done, &ctx->c_nit, &ctx->c_active, &ctx->c_task, &ctx->c_x0, &ctx->c_s0,
                      &ctx->c_noise};
    for (DevBuf *b : bufs)
        if (b->p) hipFree(b->p);
    if (ctx->h_pin) hipHostFree(ctx->h_pin);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return MP_OK;
}

int mp_set_stream(mp_ctx *ctx, void *hip_stream)
{
    DeviceGuard guard_;
    int rc = guard_.enter(ctx);
    if (rc) return rc;
    hipStream_t next = (hipStream_t)hip_stream;
    if (next == ctx->stream) return MP_OK;
    // Work queued on the old stream finishes first - unless the new stream is being captured into a graph: a
    // synchronisation is not allowed then (it would invalidate the capture), and ordering against earlier work is the
    // capturing caller's business (torch.cuda.graph waits for the side stream before it starts capturing).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (next == nullptr || hipStreamIsCapturing(next, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap == hipStreamCaptureStatusNone) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = next;
    return MP_OK;
}

int mp_use_own_stream(mp_ctx *ctx)
{
    DeviceGuard guard_;
    int rc = guard_.enter(ctx);
    if (rc) return rc;
    if (ctx->stream == ctx->own_stream) return MP_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = ctx->own_stream;
    return MP_OK;
}

int mp_synchronize(mp_ctx *ctx)
{
    DeviceGuard guard_;
    int rc = guard_.enter(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MP_OK;
}

int mp_timer_start(mp_ctx *ctx)
{
    DeviceGuard guard_;
    int rc = guard_.enter(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return MP_OK;
}

int mp_timer_stop(mp_ctx *ctx, float *elapsed_ms)
{
    DeviceGuard guard_;
    int rc = guard_.enter(ctx);
    if (rc) return rc;
    if (!elapsed_ms) return fail(ctx, MP_E_ARG, "mp_timer_stop: elapsed_ms is NULL");
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return MP_OK;
}

int mp_set_scene(mp_ctx *ctx, int32_t n_boxes, const double *cx, const double *cy, const double *cz, const double *hx,
                 const double *hy, const double *hz, const double *quat_xyzw)
{
    DeviceGuard guard_;
    int rc = guard_.enter(ctx);
    if (rc) return rc;
    if (n_boxes < 0) return fail(ctx, MP_E_ARG, "mp_set_scene: n_boxes < 0");
    if (n_boxes > 0 && (!cx || !cy || !cz || !hx || !hy || !hz)) return fail(ctx, MP_E_ARG, "mp_set_scene: NULL array");
    std::vector<double> cb((size_t)std::max(n_boxes, 1), 1.0), sb((size_t)std::max(n_boxes, 1), 0.0), rtab;
    for (int i = 0; i < n_boxes; ++i) {
        if (!(hx[i] >= 0.0 && hy[i] >= 0.0 && hz[i] >= 0.0))
            return fail(ctx, MP_E_ARG, "mp_set_scene: box %d has a negative or NaN half extent", i);
        if (quat_xyzw) {
            const double *q = quat_xyzw + 4 * (size_t)i;
            const double x = q[0], y = q[1], z = q[2], w = q[3];
            const double nn = x * x + y * y + z * z + w * w;
            if (!(std::fabs(nn - 1.0) <= 1e-9))
                return fail(ctx, MP_E_ARG, "mp_set_scene: box %d has a quaternion that is not of unit length", i);
            if (x == 0.0 && y == 0.0) {
                // R00 / R10 of the reference's quat_to_rot (geom/utils.py:3-14) for a pure-yaw quaternion
                if (!(z == 0.0)) {
                    cb[i] = 1.0 - 2.0 * (z * z);
                    sb[i] = 2.0 * (w * z);
                }
            } else {
                // general rotation: the full matrix of quat_to_rot, row major
                const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
                const double R[9] = {1 - 2 * (yy + zz), 2 * (xy - wz), 2 * (xz + wy), 2 * (xy + wz), 1 - 2 * (xx + zz), 2 * (yz - wx),
                                     2 * (xz - wy), 2 * (yz + wx), 1 - 2 * (xx + yy)};
                cb[i] = (double)(rtab.size() / 9);
                sb[i] = 2.0;
                rtab.insert(rtab.end(), R, R + 9);
            }
        }
    }
    rc = ctx->dtype == MP_F64 ? upload_scene<double>(ctx, n_boxes, cx, cy, cz, hx, hy, hz, cb.data(), sb.data(), rtab)
                              : upload_scene<float>(ctx, n_boxes, cx, cy, cz, hx, hy, hz, cb.data(), sb.data(), rtab);
    if (rc) return rc;
    ctx->n_boxes = n_boxes;
    ctx->have_scene = true;
    ctx->have_problem = false;  // a problem's box ranges and bound tables belong to the scene it was set on
    return MP_OK;
}

int mp_build_basis(int32_t n_ctrl, int32_t degree, int32_t n_samples, double *M)
{
    if (!M) return MP_E_ARG;
    return build_basis(n_ctrl, degree, n_samples, M);
}

int mp_set_problem(mp_ctx *ctx, const mp_config *cfg, int32_t n_tasks, const mp_task *tasks)
{
    DeviceGuard guard_;
    int rc = guard_.enter(ctx);
    if (rc) return rc;
    if (!cfg || !tasks || n_tasks < 1) return fail(ctx, MP_E_ARG, "mp_set_problem: NULL cfg/tasks or n_tasks < 1");
    if (!ctx->have_scene) return fail(ctx, MP_E_STATE, "mp_set_problem: call mp_set_scene first");
    const mp_config &c = *cfg;
    if (c.n_samples < 2 || c.n_samples > MP_MAX_SAMPLES) return fail(ctx, MP_E_ARG, "n_samples out of range");
    if (c.n_ctrl < 4 || c.n_ctrl > MP_MAX_CTRL) return fail(ctx, MP_E_ARG, "n_ctrl must be in 4..%d", MP_MAX_CTRL);
    if (c.n_yaw_ctrl < 2 || c.n_yaw_ctrl > MP_MAX_CTRL) return fail(ctx, MP_E_ARG, "n_yaw_ctrl must be in 2..%d", MP_MAX_CTRL);
    if (c.n_free < 1 || c.first_free < 1 || c.first_free + c.n_free > c.n_ctrl - 1)
        return fail(ctx, MP_E_ARG, "free waypoint rows [%d,%d) must lie strictly inside the %d waypoints", c.first_free,
                    c.first_free + c.n_free, c.n_ctrl);
    if (c.n_yaw_free != c.n_yaw_ctrl - 2) return fail(ctx, MP_E_ARG, "n_yaw_free must equal n_yaw_ctrl - 2");
    if (c.model != MP_MODEL_POINT && c.model != MP_MODEL_BOX) return fail(ctx, MP_E_ARG, "unknown collision model");
    if (c.n_safe < 1 || c.n_safe > c.n_samples) return fail(ctx, MP_E_ARG, "n_safe must be in 1..n_samples");
    if (!c.idx_safe || !c.approach_mask || !c.t_sched) return fail(ctx, MP_E_ARG, "idx_safe / approach_mask / t_sched is NULL");
    for (int k = 0; k < c.n_safe; ++k) {
        if (c.idx_safe[k] < 0 || c.idx_safe[k] >= c.n_samples) return fail(ctx, MP_E_ARG, "idx_safe[%d] out of range", k);
        if (k > 0 && c.idx_safe[k] <= c.idx_safe[k - 1]) return fail(ctx, MP_E_ARG, "idx_safe must be strictly increasing");
    }
    if (c.keep_n < 0 || c.tail_start < 0 || c.goal_tail_n < 0) return fail(ctx, MP_E_ARG, "negative window");
    for (int i = 0; i < n_tasks; ++i) {
        const mp_task &t = tasks[i];
        if (t.box_begin < 0 || t.box_count < 0 || t.box_begin + t.box_count > ctx->n_boxes)
            return fail(ctx, MP_E_ARG, "task %d: boxes [%d,%d) outside the scene (%d boxes)", i, t.box_begin,
                        t.box_begin + t.box_count, ctx->n_boxes);
    }
    ctx->have_problem = false;
    ctx->cfg = c;
    ctx->n_tasks = n_tasks;
    ctx->max_task_boxes = 0;
    for (int i = 0; i < n_tasks; ++i) ctx->max_task_boxes = std::max(ctx->max_task_boxes, (int)tasks[i].box_count);
    ctx->basis.assign((size_t)c.n_samples * c.n_ctrl, 0.0);
    if (c.basis) std::memcpy(ctx->basis.data(), c.basis, ctx->basis.size() * sizeof(double));
    else if ((rc = build_basis(c.n_ctrl, 3, c.n_samples, ctx->basis.data()))) return fail(ctx, rc, "basis construction failed");
    ctx->basis_yaw.assign((size_t)c.n_samples * c.n_yaw_ctrl, 0.0);
    if (c.basis_yaw) std::memcpy(ctx->basis_yaw.data(), c.basis_yaw, ctx->basis_yaw.size() * sizeof(double));
    else if ((rc = build_basis(c.n_yaw_ctrl, std::min(3, c.n_yaw_ctrl - 1), c.n_samples, ctx->basis_yaw.data())))
        return fail(ctx, rc, "yaw basis construction failed");
    if ((rc = upload(ctx, ctx->d_idx, c.idx_safe, (size_t)c.n_safe * sizeof(int32_t)))) return rc;
    std::vector<int32_t> inv((size_t)c.n_samples, -1);   // spline sample -> collision sample taken there
    for (int k = 0; k < c.n_safe; ++k) inv[(size_t)c.idx_safe[k]] = k;
    if ((rc = upload(ctx, ctx->d_inv, inv.data(), inv.size() * sizeof(int32_t)))) return rc;
    ctx->idx_identity = 1;
    for (int k = 0; k < c.n_safe; ++k)
        if (c.idx_safe[k] != k) { ctx->idx_identity = 0; break; }
    if ((rc = upload(ctx, ctx->d_mask, c.approach_mask, (size_t)c.n_safe))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    rc = ctx->dtype == MP_F64 ? upload_problem<double>(ctx, tasks) : upload_problem<float>(ctx, tasks);
    if (rc) return rc;
    // the copied config must not keep caller pointers
    ctx->cfg.idx_safe = nullptr; ctx->cfg.approach_mask = nullptr; ctx->cfg.basis = nullptr;
    ctx->cfg.basis_yaw = nullptr; ctx->cfg.t_sched = nullptr;
    ctx->have_problem = true;
    return MP_OK;
}

int mp_get_basis(mp_ctx *ctx, double *M, double *M_yaw)
{
    if (!ctx) return MP_E_ARG;
    if (!ctx->have_problem) return fail(ctx, MP_E_STATE, "mp_get_basis: no problem set");
    if (M) std::memcpy(M, ctx->basis.data(), ctx->basis.size() * sizeof(double));
    if (M_yaw) std::memcpy(M_yaw, ctx->basis_yaw.data(), ctx->basis_yaw.size() * sizeof(double));
    return MP_OK;
}

int mp_set_cands_per_task(mp_ctx *ctx, int64_t cands_per_task)
{
    if (!ctx) return MP_E_ARG;
    if (cands_per_task < 0) return fail(ctx, MP_E_ARG, "cands_per_task < 0");
    ctx->cands_per_task = cands_per_task;
    return MP_OK;
}

int mp_eval(mp_ctx *ctx, int64_t n, const void *X, const int32_t *task_of, void *cost, void *terms, void *grad,
            void *min_clear, uint8_t *flags, void *dist, int mem)
{
    DeviceGuard guard_;
    int rc = guard_.enter(ctx);
    if (rc) return rc;
    if (!ctx->have_problem) return fail(ctx, MP_E_STATE, "mp_eval: call mp_set_scene and mp_set_problem first");
    if (n < 0 || (n > 0 && (!X || !cost))) return fail(ctx, MP_E_ARG, "mp_eval: X and cost are required");
    if (n == 0) return MP_OK;
    if (mem != MP_MEM_HOST && mem != MP_MEM_DEVICE) return fail(ctx, MP_E_ARG, "mp_eval: unknown mem kind");
    const mp_config &c = ctx->cfg;
    const size_t es = esize(ctx);
    const size_t D = (size_t)(3 * c.n_free + c.n_yaw_free);
    if (!task_of && ctx->n_tasks > 1) {
        if (ctx->cands_per_task <= 0) return fail(ctx, MP_E_ARG, "mp_eval: several tasks need task_of or mp_set_cands_per_task");
        if ((n + ctx->cands_per_task - 1) / ctx->cands_per_task > ctx->n_tasks)
            return fail(ctx, MP_E_ARG, "mp_eval: n exceeds n_tasks * cands_per_task");
    }
    if (!task_of && ctx->n_tasks == 1 && ctx->cands_per_task > 0 && n > ctx->cands_per_task)
        return fail(ctx, MP_E_ARG, "mp_eval: n exceeds n_tasks * cands_per_task");

    if (mem == MP_MEM_DEVICE) {
        return ctx->dtype == MP_F64 ? launch_eval<double>(ctx, n, X, task_of, cost, terms, grad, min_clear, flags, dist)
                                    : launch_eval<float>(ctx, n, X, task_of, cost, terms, grad, min_clear, flags, dist);
    }
    // host pointers: stage through device buffers owned by the context
    if (task_of)
        for (int64_t i = 0; i < n; ++i)
            if (task_of[i] < 0 || task_of[i] >= ctx->n_tasks) return fail(ctx, MP_E_ARG, "mp_eval: task_of[%lld] out of range", (long long)i);
    {
        // Small calls (one x from SciPy's Powell, one CEM population): the kernel reads X from and writes its results
        // to pinned host memory directly, so the call is one launch and one stream synchronisation instead of up to
        // eight copy commands (measured: 66-100 us -> see DESIGN.md 7).
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t bX = up((size_t)n * D * es), bT = task_of ? up((size_t)n * 4) : 0, bC = up((size_t)n * es);
        const size_t bTerms = terms ? up((size_t)n * MP_NTERMS * es) : 0, bG = grad ? up((size_t)n * D * es) : 0;
        const size_t bM = min_clear ? up((size_t)n * es) : 0, bF = flags ? up((size_t)n) : 0;
        const size_t bD = dist ? up((size_t)n * c.n_safe * es) : 0;
        const size_t total = bX + bT + bC + bTerms + bG + bM + bF + bD;
        if (total <= kPinnedCallBytes) {
            if (total > ctx->h_pin_cap) {
                if (ctx->h_pin) HIP_TRY(ctx, hipHostFree(ctx->h_pin));
                ctx->h_pin = nullptr; ctx->h_pin_cap = 0;
                HIP_TRY(ctx, hipHostMalloc(&ctx->h_pin, kPinnedCallBytes, hipHostMallocDefault));
                ctx->h_pin_cap = kPinnedCallBytes;
            }
            unsigned char *q = (unsigned char *)ctx->h_pin;
            unsigned char *pX = q; q += bX;
            unsigned char *pT = q; q += bT;
            unsigned char *pC = q; q += bC;
            unsigned char *pTerms = q; q += bTerms;
            unsigned char *pG = q; q += bG;
            unsigned char *pM = q; q += bM;
            unsigned char *pF = q; q += bF;
            unsigned char *pD = q;
            std::memcpy(pX, X, (size_t)n * D * es);
            if (task_of) std::memcpy(pT, task_of, (size_t)n * 4);
            const int32_t *zt = task_of ? (const int32_t *)pT : nullptr;
            void *zterms = terms ? pTerms : nullptr, *zg = grad ? pG : nullptr, *zm = min_clear ? pM : nullptr;
            void *zd = dist ? pD : nullptr;
            uint8_t *zf = flags ? pF : nullptr;
            rc = ctx->dtype == MP_F64 ? launch_eval<double>(ctx, n, pX, zt, pC, zterms, zg, zm, zf, zd)
                                      : launch_eval<float>(ctx, n, pX, zt, pC, zterms, zg, zm, zf, zd);
            if (rc) return rc;
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            std::memcpy(cost, pC, (size_t)n * es);
            if (terms) std::memcpy(terms, pTerms, (size_t)n * MP_NTERMS * es);
            if (grad) std::memcpy(grad, pG, (size_t)n * D * es);
            if (min_clear) std::memcpy(min_clear, pM, (size_t)n * es);
            if (flags) std::memcpy(flags, pF, (size_t)n);
            if (dist) std::memcpy(dist, pD, (size_t)n * c.n_safe * es);
            return MP_OK;
        }
    }
    if ((rc = upload(ctx, ctx->s_X, X, (size_t)n * D * es))) return rc;
    if (task_of && (rc = upload(ctx, ctx->s_taskof, task_of, (size_t)n * sizeof(int32_t)))) return rc;
    if ((rc = reserve(ctx, ctx->s_cost, (size_t)n * es))) return rc;
    if (terms && (rc = reserve(ctx, ctx->s_terms, (size_t)n * MP_NTERMS * es))) return rc;
    if (grad && (rc = reserve(ctx, ctx->s_grad, (size_t)n * D * es))) return rc;
    if (min_clear && (rc = reserve(ctx, ctx->s_minc, (size_t)n * es))) return rc;
    if (flags && (rc = reserve(ctx, ctx->s_flags, (size_t)n))) return rc;
    if (dist && (rc = reserve(ctx, ctx->s_dist, (size_t)n * c.n_safe * es))) return rc;
    const int32_t *d_taskof = task_of ? (const int32_t *)ctx->s_taskof.p : nullptr;
    void *d_terms = terms ? ctx->s_terms.p : nullptr, *d_grad = grad ? ctx->s_grad.p : nullptr;
    void *d_minc = min_clear ? ctx->s_minc.p : nullptr, *d_dist = dist ? ctx->s_dist.p : nullptr;
    uint8_t *d_flags = flags ? (uint8_t *)ctx->s_flags.p : nullptr;
    rc = ctx->dtype == MP_F64
             ? launch_eval<double>(ctx, n, ctx->s_X.p, d_taskof, ctx->s_cost.p, d_terms, d_grad, d_minc, d_flags, d_dist)
             : launch_eval<float>(ctx, n, ctx->s_X.p, d_taskof, ctx->s_cost.p, d_terms, d_grad, d_minc, d_flags, d_dist);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(cost, ctx->s_cost.p, (size_t)n * es, hipMemcpyDeviceToHost, ctx->stream));
    if (terms) HIP_TRY(ctx, hipMemcpyAsync(terms, d_terms, (size_t)n * MP_NTERMS * es, hipMemcpyDeviceToHost, ctx->stream));
    if (grad) HIP_TRY(ctx, hipMemcpyAsync(grad, d_grad, (size_t)n * D * es, hipMemcpyDeviceToHost, ctx->stream));
    if (min_clear) HIP_TRY(ctx, hipMemcpyAsync(min_clear, d_minc, (size_t)n * es, hipMemcpyDeviceToHost, ctx->stream));
    if (flags) HIP_TRY(ctx, hipMemcpyAsync(flags, d_flags, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (dist) HIP_TRY(ctx, hipMemcpyAsync(dist, d_dist, (size_t)n * c.n_safe * es, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MP_OK;
}

int mp_eval_noalloc(mp_ctx *ctx, int64_t n, const void *X, const int32_t *task_of, void *cost, void *grad, void *min_clear,
                    uint8_t *flags)
{
    DeviceGuard guard_;
    int rc = guard_.enter(ctx);
    if (rc) return rc;
    if (!ctx->have_problem) return fail(ctx, MP_E_STATE, "mp_eval_noalloc: call mp_set_scene and mp_set_problem first");
    if (n < 0 || (n > 0 && (!X || !cost))) return fail(ctx, MP_E_ARG, "mp_eval_noalloc: X and cost are required");
    if (n == 0) return MP_OK;
    const mp_config &c = ctx->cfg;
    if (c.model != MP_MODEL_BOX)
        return fail(ctx, MP_E_UNSUPPORTED, "mp_eval_noalloc: only the box model has kernels with a scratch table; for the point "
                    "model mp_eval(MP_MEM_DEVICE) never allocates");
    if (!task_of && ctx->n_tasks > 1) {
        if (ctx->cands_per_task <= 0) return fail(ctx, MP_E_ARG, "mp_eval_noalloc: several tasks need task_of or mp_set_cands_per_task");
        if ((n + ctx->cands_per_task - 1) / ctx->cands_per_task > ctx->n_tasks)
            return fail(ctx, MP_E_ARG, "mp_eval_noalloc: n exceeds n_tasks * cands_per_task");
    }
    if (!task_of && ctx->n_tasks == 1 && ctx->cands_per_task > 0 && n > ctx->cands_per_task)
        return fail(ctx, MP_E_ARG, "mp_eval_noalloc: n exceeds n_tasks * cands_per_task");
    void *ptr = ctx->device < 64 ? g_scratch_ptr[ctx->device].load(std::memory_order_acquire) : nullptr;
    if (!ptr) return fail(ctx, MP_E_STATE, "mp_eval_noalloc: no scratch address for device %d", ctx->device);
    const size_t es = esize(ctx), samples = (size_t)n * (size_t)c.n_safe;
    const size_t need = noalloc_arg_offset(samples * es) + samples * sizeof(int);
    if (need > (size_t)MP_NOALLOC_SCRATCH_BYTES)
        return fail(ctx, MP_E_ARG, "mp_eval_noalloc: %lld candidates x %d collision samples need %zu bytes of scratch, the static "
                    "array holds %zu (MP_NOALLOC_SCRATCH_BYTES): split the batch or call mp_eval", (long long)n, c.n_safe, need,
                    (size_t)MP_NOALLOC_SCRATCH_BYTES);
    // Nothing in the tables has to start at zero: clearance_kernel writes every (candidate, sample) entry that
    // eval_kernel reads afterwards on the same stream; entries of the previous call beyond n x S' are never read.
    ScratchOrder &ord = g_scratch_order[ctx->device];
    std::lock_guard<std::mutex> lock(ord.m);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (ctx->stream == nullptr || hipStreamIsCapturing(ctx->stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (!capturing && ord.recorded && ord.last != ctx->stream)
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ord.done, 0));   // the previous user of this GPU's tables, on another stream
    rc = ctx->dtype == MP_F64
             ? launch_eval<double>(ctx, n, X, task_of, cost, nullptr, grad, min_clear, flags, nullptr, 0, nullptr, nullptr, ptr)
             : launch_eval<float>(ctx, n, X, task_of, cost, nullptr, grad, min_clear, flags, nullptr, 0, nullptr, nullptr, ptr);
    if (rc) return rc;
    if (!capturing) {
        HIP_TRY(ctx, hipEventRecord(ord.done, ctx->stream));
        ord.last = ctx->stream;
        ord.recorded = true;
    }
    return MP_OK;
}

// ---- the one collective of a sharded job, for callers without torch.distributed ------------------------
// RCCL is not linked: its ncclAllGather is looked up at the first call, in the process first (the caller made the
// communicator, so its RCCL is loaded), then by name.  ncclDataType_t ncclUint8 = 1, ncclResult_t ncclSuccess = 0.
namespace {
typedef int (*nccl_allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
typedef const char *(*nccl_errstr_fn)(int);
std::mutex g_rccl_mutex;
nccl_allgather_fn g_nccl_allgather = nullptr;
nccl_errstr_fn g_nccl_errstr = nullptr;
bool find_rccl()
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_nccl_allgather) return true;
    void *sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
    void *es = sym ? dlsym(RTLD_DEFAULT, "ncclGetErrorString") : nullptr;
    if (!sym) {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        void *h = nullptr;
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;   // a copy the process has loaded already
        for (size_t i = 0; !h && i < sizeof names / sizeof *names; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (h) { sym = dlsym(h, "ncclAllGather"); es = dlsym(h, "ncclGetErrorString"); }
    }
    g_nccl_errstr = (nccl_errstr_fn)es;
    g_nccl_allgather = (nccl_allgather_fn)sym;
    return g_nccl_allgather != nullptr;
}
}  // namespace

int mp_allgather_results(mp_ctx *ctx, void *nccl_comm, const void *send, void *recv, size_t bytes_per_rank)
{
    DeviceGuard guard_;
    int rc = guard_.enter(ctx);
    if (rc) return rc;
    if (!nccl_comm || !send || !recv) return fail(ctx, MP_E_ARG, "mp_allgather_results: NULL argument");
    if (bytes_per_rank == 0) return MP_OK;
    if (!find_rccl()) {
        const char *why = dlerror();
        return fail(ctx, MP_E_UNSUPPORTED, "mp_allgather_results: no RCCL in this process (ncclAllGather not found; %s)", why ? why : "no loader message");
    }
    const int r = g_nccl_allgather(send, recv, bytes_per_rank, 1 /* ncclUint8 */, nccl_comm, ctx->stream);
    if (r != 0) return fail(ctx, MP_E_HIP, "ncclAllGather failed: %s", g_nccl_errstr ? g_nccl_errstr(r) : "unknown RCCL error");
    return MP_OK;
}

static int sdf_common(mp_ctx *ctx, int64_t n, int32_t box_begin, int32_t box_count, int mem, const char *who)
{
    if (!ctx->have_scene) return fail(ctx, MP_E_STATE, "%s: call mp_set_scene first", who);
    if (n < 0) return fail(ctx, MP_E_ARG, "%s: n < 0", who);
    if (box_begin < 0 || box_count < 0 || box_begin + box_count > ctx->n_boxes)
        return fail(ctx, MP_E_ARG, "%s: boxes [%d,%d) outside the scene (%d boxes)", who, box_begin, box_begin + box_count,
                    ctx->n_boxes);
    if (mem != MP_MEM_HOST && mem != MP_MEM_DEVICE) return fail(ctx, MP_E_ARG, "%s: unknown mem kind", who);
    return MP_OK;
}

int mp_point_sdf(mp_ctx *ctx, int64_t n, const void *pts, double point_radius, int32_t box_begin, int32_t box_count,
                 void *out, int mem)
{
    DeviceGuard guard_;
    int rc = guard_.enter(ctx);
    if (rc) return rc;
    if ((rc = sdf_common(ctx, n, box_begin, box_count, mem, "mp_point_sdf"))) return rc;
    if (n == 0) return MP_OK;
    if (!pts || !out) return fail(ctx, MP_E_ARG, "mp_point_sdf: NULL pointer");
    const size_t es = esize(ctx);
    const void *d_pts = pts;
    void *d_out = out;
    if (mem == MP_MEM_HOST) {
        if ((rc = upload(ctx, ctx->s_aux0, pts, (size_t)n * 3 * es))) return rc;
        if ((rc = reserve(ctx, ctx->s_aux1, (size_t)n * es))) return rc;
        d_pts = ctx->s_aux0.p;
        d_out = ctx->s_aux1.p;
    }
    int blocks = (int)std::min<long long>((n + 255) / 256, (long long)ctx->num_cu * 8);
    if (ctx->dtype == MP_F64)
        hipLaunchKernelGGL(mp::point_sdf_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, (long long)n,
                           (const double *)d_pts, (const mp::BoxRec<double> *)ctx->boxes.p + box_begin, box_count,
                           point_radius, ctx->has_rot, (const double *)ctx->tilt.p, (double *)d_out);
    else
        hipLaunchKernelGGL(mp::point_sdf_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream, (long long)n,
                           (const float *)d_pts, (const mp::BoxRec<float> *)ctx->boxes.p + box_begin, box_count,
                           (float)point_radius, ctx->has_rot, (const float *)ctx->tilt.p, (float *)d_out);
    HIP_TRY(ctx, hipGetLastError());
    if (mem == MP_MEM_HOST) {
        HIP_TRY(ctx, hipMemcpyAsync(out, d_out, (size_t)n * es, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MP_OK;
}

int mp_box_sdf(mp_ctx *ctx, int64_t n, const void *pos, const void *yaw_deg, const double size[3], int32_t box_begin,
               int32_t box_count, void *out, int mem)
{
    DeviceGuard guard_;
    int rc = guard_.enter(ctx);
    if (rc) return rc;
    if ((rc = sdf_common(ctx, n, box_begin, box_count, mem, "mp_box_sdf"))) return rc;
    if (n == 0) return MP_OK;
    if (!pos || !yaw_deg || !size || !out) return fail(ctx, MP_E_ARG, "mp_box_sdf: NULL pointer");
    const size_t es = esize(ctx);
    const void *d_pos = pos, *d_yaw = yaw_deg;
    void *d_out = out;
    if (mem == MP_MEM_HOST) {
        // pos | yaw packed in one staging buffer
        if ((rc = reserve(ctx, ctx->s_aux0, (size_t)n * 4 * es))) return rc;
        if ((rc = reserve(ctx, ctx->s_aux1, (size_t)n * es))) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->s_aux0.p, pos, (size_t)n * 3 * es, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync((char *)ctx->s_aux0.p + (size_t)n * 3 * es, yaw_deg, (size_t)n * es,
                                    hipMemcpyHostToDevice, ctx->stream));
        d_pos = ctx->s_aux0.p;
        d_yaw = (char *)ctx->s_aux0.p + (size_t)n * 3 * es;
        d_out = ctx->s_aux1.p;
    }
    int blocks = (int)std::min<long long>((n + 255) / 256, (long long)ctx->num_cu * 8);
    const double hx = 0.5 * size[0], hy = 0.5 * size[1], hz = 0.5 * size[2];
    if (ctx->dtype == MP_F64)
        hipLaunchKernelGGL(mp::box_sdf_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, (long long)n,
                           (const double *)d_pos, (const double *)d_yaw,
                           (const mp::BoxRec<double> *)ctx->boxes.p + box_begin, box_count, hx, hy, hz,
                           ctx->has_rot == 2 ? (const double *)ctx->tilt.p : nullptr, (double *)d_out, mp::trig_table<double>());
    else
        hipLaunchKernelGGL(mp::box_sdf_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream, (long long)n,
                           (const float *)d_pos, (const float *)d_yaw,
                           (const mp::BoxRec<float> *)ctx->boxes.p + box_begin, box_count, (float)hx, (float)hy,
                           (float)hz, ctx->has_rot == 2 ? (const float *)ctx->tilt.p : nullptr, (float *)d_out, mp::trig_table<float>());
    HIP_TRY(ctx, hipGetLastError());
    if (mem == MP_MEM_HOST) {
        HIP_TRY(ctx, hipMemcpyAsync(out, d_out, (size_t)n * es, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MP_OK;
}

int mp_cem_solve(mp_ctx *ctx, int64_t n_inst, const int32_t *task_of_inst, const void *x0, const void *sigma0,
                 const mp_cem_options *opt, const void *noise, void *best_x, void *best_val, int32_t *nit,
                 void *best_min_clear, uint8_t *best_flags, int mem)
{
    DeviceGuard guard_;
    int rc = guard_.enter(ctx);
    if (rc) return rc;
    if (!ctx->have_problem) return fail(ctx, MP_E_STATE, "mp_cem_solve: call mp_set_scene and mp_set_problem first");
    if (n_inst < 0) return fail(ctx, MP_E_ARG, "mp_cem_solve: n_inst < 0");
    if (n_inst == 0) return MP_OK;
    if (!x0 || !sigma0 || !opt || !best_x || !best_val) return fail(ctx, MP_E_ARG, "mp_cem_solve: NULL argument");
    if (mem != MP_MEM_HOST && mem != MP_MEM_DEVICE) return fail(ctx, MP_E_ARG, "mp_cem_solve: unknown mem kind");
    if (opt->population_size < 2 || opt->population_size > 1024) return fail(ctx, MP_E_ARG, "population_size must be in 2..1024");
    if (opt->max_iter < 1) return fail(ctx, MP_E_ARG, "max_iter must be >= 1");
    if (!(opt->elite_frac > 0.0 && opt->elite_frac <= 1.0)) return fail(ctx, MP_E_ARG, "elite_frac must be in (0, 1]");
    if (!(opt->alpha >= 0.0 && opt->alpha <= 1.0) || !(opt->min_sigma >= 0.0)) return fail(ctx, MP_E_ARG, "alpha in [0,1], min_sigma >= 0");
    const mp_config &c = ctx->cfg;
    const size_t es = esize(ctx), D = (size_t)(3 * c.n_free + c.n_yaw_free);
    // instance -> task map on the device
    std::vector<int32_t> tmap((size_t)n_inst);
    const long long per = (n_inst + ctx->n_tasks - 1) / ctx->n_tasks;
    for (int64_t i = 0; i < n_inst; ++i) {
        // task_of_inst is a host pointer in both mem modes (it is bookkeeping, not data)
        const long long t = task_of_inst ? task_of_inst[i] : i / per;
        if (t < 0 || t >= ctx->n_tasks) return fail(ctx, MP_E_ARG, "mp_cem_solve: task of instance %lld out of range", (long long)i);
        tmap[(size_t)i] = (int32_t)t;
    }
    if ((rc = upload(ctx, ctx->c_task, tmap.data(), tmap.size() * sizeof(int32_t)))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const void *d_x0 = x0, *d_s0 = sigma0, *d_noise = noise;
    if (mem == MP_MEM_HOST) {
        if ((rc = upload(ctx, ctx->c_x0, x0, (size_t)n_inst * D * es))) return rc;
        if ((rc = upload(ctx, ctx->c_s0, sigma0, (size_t)n_inst * D * es))) return rc;
        d_x0 = ctx->c_x0.p; d_s0 = ctx->c_s0.p;
        if (noise) {
            if ((rc = upload(ctx, ctx->c_noise, noise, (size_t)opt->max_iter * n_inst * opt->population_size * D * es))) return rc;
            d_noise = ctx->c_noise.p;
        }
    }
    if (ctx->dtype == MP_F64)
        return cem_solve<double>(ctx, n_inst, (const int32_t *)ctx->c_task.p, (const double *)d_x0, (const double *)d_s0, *opt,
                                 (const double *)d_noise, (double *)best_x, (double *)best_val, nit, (double *)best_min_clear,
                                 best_flags, mem);
    return cem_solve<float>(ctx, n_inst, (const int32_t *)ctx->c_task.p, (const float *)d_x0, (const float *)d_s0, *opt,
                            (const float *)d_noise, (float *)best_x, (float *)best_val, nit, (float *)best_min_clear, best_flags,
                            mem);
}

}  // extern "C"

#ifdef MP_STATS
// investigation build only (-DMP_STATS): read and reset the work counters of the clearance phase
extern "C" int mp_debug_stats(unsigned long long *out8)
{
    if (hipDeviceSynchronize() != hipSuccess) return MP_E_HIP;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(mp::g_stats), 8 * sizeof(unsigned long long)) != hipSuccess) return MP_E_HIP;
    unsigned long long z[8] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mp::g_stats), z, sizeof(z)) != hipSuccess) return MP_E_HIP;
    return MP_OK;
}

#endif

#ifdef MP_PHASES
extern "C" int mp_debug_phases(unsigned long long *out16)
{
    if (hipDeviceSynchronize() != hipSuccess) return MP_E_HIP;
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(mp::g_phase), 16 * sizeof(unsigned long long)) != hipSuccess) return MP_E_HIP;
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mp::g_phase), z, sizeof(z)) != hipSuccess) return MP_E_HIP;
    return MP_OK;
}
#endif
