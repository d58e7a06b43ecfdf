// mp_device.hpp — device-side math of the trajectory evaluation kernels (gfx950).
//
// Arithmetic contract (DESIGN.md §"Numerics"): IEEE + - * / sqrt, fma exactly where
// written (mp_fma), no other contraction (the translation unit is compiled with
// -ffp-contract=off), own sin/cos.  The CPU oracle follows the same contract, which is
// what makes the collision / feasibility flags bit-exact between the two.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace mp {

template <typename T> struct Lim;
template <> struct Lim<double> {
    __device__ static double inf() { return __builtin_huge_val(); }
    __device__ static double pio2_hi() { return 1.57079632679489655800e+00; }
    __device__ static double pio2_lo() { return 6.12323399573676603587e-17; }
};
template <> struct Lim<float> {
    __device__ static float inf() { return __builtin_huge_valf(); }
    __device__ static float pio2_hi() { return 1.57079637050628662109e+00f; }
    __device__ static float pio2_lo() { return -4.37113882867379288655e-08f; }
};

__device__ __forceinline__ double mp_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float mp_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double mp_abs(double a) { return __builtin_fabs(a); }
__device__ __forceinline__ float mp_abs(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ double mp_sqrt(double a) { return __builtin_sqrt(a); }
__device__ __forceinline__ float mp_sqrt(float a) { return __builtin_sqrtf(a); }
__device__ __forceinline__ double mp_rint(double a) { return __builtin_rint(a); }
__device__ __forceinline__ float mp_rint(float a) { return __builtin_rintf(a); }
// max/min for finite operands (±0 may differ from a compare-select; callers only square
// or compare the result)
__device__ __forceinline__ double mp_max(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float mp_max(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double mp_min(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float mp_min(float a, float b) { return __builtin_fminf(a, b); }

template <typename T> __device__ __forceinline__ T sgn(T v) { return v < T(0) ? T(-1) : T(1); }
// sqrt(q) and 1/sqrt(q) together, from the hardware reciprocal-square-root estimate and Newton steps
// (coupled Goldschmidt iteration): ~47 SIMD cycles per wave in fp64 against 97 for the correctly rounded sqrt plus
// 71 for a division (tools/micro/f64_ops.hip).  Results are within ~1 ulp, NOT correctly rounded: used only where
// the parity bar is a tolerance (cost terms, gradient), never on the path to the clearances and flags.
// Zero, denormal-range and non-finite inputs take the exact path (wave-uniform branch, rare).
__device__ __forceinline__ void sqrt_rsqrt(double q, double &s, double &inv)
{
    if (__builtin_amdgcn_ballot_w64(!(q > 1e-280 && q < 1e280))) {
        s = __builtin_sqrt(q);
        inv = 1.0 / s;
        return;
    }
    const double y = __builtin_amdgcn_rsq(q);
    double g = q * y, h = 0.5 * y;
    double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
    e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
    s = g; inv = h + h;
}
__device__ __forceinline__ void sqrt_rsqrt(float q, float &s, float &inv)
{
    if (__builtin_amdgcn_ballot_w64(!(q > 1e-30f && q < 1e30f))) {
        s = __builtin_sqrtf(q);
        inv = 1.0f / s;
        return;
    }
    const float y = __builtin_amdgcn_rsqf(q);
    float g = q * y, h = 0.5f * y;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, e, g); h = __builtin_fmaf(h, e, h);
    s = g; inv = h + h;
}
// 1/b to ~1 ulp (reciprocal estimate + Newton); same restriction as sqrt_rsqrt
__device__ __forceinline__ double fast_rcp(double b)
{
    double x = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, x, 1.0);
    x = __builtin_fma(x, e, x);
    e = __builtin_fma(-b, x, 1.0);
    return __builtin_fma(x, e, x);
}
__device__ __forceinline__ float fast_rcp(float b)
{
    float x = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, x, 1.0f);
    return __builtin_fmaf(x, e, x);
}

// np.maximum(0, v): NaN-propagating
template <typename T> __device__ __forceinline__ T npmax0(T v) { return !(v <= T(0)) ? v : T(0); }

// ---- sin/cos of the half yaw angle -------------------------------------------------
// Cody-Waite reduction by pi/2 (two fma steps) + degree-13/12 minimax kernels on
// [-pi/4, pi/4]; the coefficient set is the classic fdlibm one.  Same op sequence as the
// oracle's orc_sincos.
// The 16 constants travel as a table in the kernel arguments (TrigTab, filled by trig_table()): a 64-bit literal
// cannot be an operand of a VALU instruction, so written as literals the compiler kept all of them in VGPR pairs for
// the whole kernel (24+ vector registers of every fp64 kernel); read from the argument segment they are scalar
// operands of the fma chain and live only while the chain runs.
template <typename T> struct TrigTab {
    T s[6], c[6];
    T two_over_pi, pio2_hi, pio2_lo, deg2rad;
};
template <typename T> __host__ __device__ inline TrigTab<T> trig_table()
{
    TrigTab<T> t;
    t.s[0] = T(-1.66666666666666324348e-01); t.s[1] = T(8.33333333332248946124e-03); t.s[2] = T(-1.98412698298579493134e-04);
    t.s[3] = T(2.75573137070700676789e-06); t.s[4] = T(-2.50507602534068634195e-08); t.s[5] = T(1.58969099521155010221e-10);
    t.c[0] = T(4.16666666666666019037e-02); t.c[1] = T(-1.38888888888741095749e-03); t.c[2] = T(2.48015872894767294178e-05);
    t.c[3] = T(-2.75573143513906633035e-07); t.c[4] = T(2.08757232129817482790e-09); t.c[5] = T(-1.13596475577881948265e-11);
    t.two_over_pi = T(0.63661977236758134308);
    t.pio2_hi = sizeof(T) == 8 ? T(1.57079632679489655800e+00) : T(1.57079637050628662109e+00);
    t.pio2_lo = sizeof(T) == 8 ? T(6.12323399573676603587e-17) : T(-4.37113882867379288655e-08);
    t.deg2rad = T(0.017453292519943295);
    return t;
}

template <typename T, typename TG> __device__ __forceinline__ void sincos_reduced(T h, const TG &g, T &sn, T &cs)
{
    T k = mp_rint(h * g.two_over_pi);
    T r = mp_fma(-k, g.pio2_hi, h);
    r = mp_fma(-k, g.pio2_lo, r);
    int q = (int)((long long)k) & 3;
    T z = r * r;
    T ps = mp_fma(z, g.s[5], g.s[4]);
    ps = mp_fma(z, ps, g.s[3]);
    ps = mp_fma(z, ps, g.s[2]);
    ps = mp_fma(z, ps, g.s[1]);
    ps = mp_fma(z, ps, g.s[0]);
    T sr = mp_fma(r * z, ps, r);
    T pc = mp_fma(z, g.c[5], g.c[4]);
    pc = mp_fma(z, pc, g.c[3]);
    pc = mp_fma(z, pc, g.c[2]);
    pc = mp_fma(z, pc, g.c[1]);
    pc = mp_fma(z, pc, g.c[0]);
    T cr = mp_fma(z * z, pc, mp_fma(T(-0.5), z, T(1.0)));
    T a = (q & 1) ? cr : sr;  // sin candidates
    T b = (q & 1) ? sr : cr;  // cos candidates
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}

// yaw in degrees -> (cos, sin) composed like yaw_deg_to_quat + quat_to_rot
// (geom/spline_opt.py:37-40, geom/utils.py:3-14).
template <typename T, typename TG> __device__ __forceinline__ void yaw_rot(T yaw_deg, const TG &g, T &c, T &s)
{
    T h = T(0.5) * (yaw_deg * g.deg2rad);
    T zq, wq;
    sincos_reduced(h, g, zq, wq);
    c = T(1.0) - T(2.0) * (zq * zq);
    s = T(2.0) * (wq * zq);
}

// ---- box model -----------------------------------------------------------------------
template <typename T> struct Pose {
    T px, py, pz, c, s;
    T ex, ey, u1x, u1y, u2x, u2y;
};

template <typename T>
__device__ __forceinline__ void pose_setup(Pose<T> &p, T px, T py, T pz, T c, T s, T ax, T ay)
{
    p.px = px; p.py = py; p.pz = pz; p.c = c; p.s = s;
    T ac = mp_abs(c), as = mp_abs(s);
    p.ex = mp_fma(as, ay, ac * ax);
    p.ey = mp_fma(ac, ay, as * ax);
    p.u1x = mp_fma(-s, ay, c * ax);
    p.u1y = mp_fma(c, ay, s * ax);
    p.u2x = mp_fma(s, ay, c * ax);
    p.u2y = mp_fma(-c, ay, s * ax);
}

template <typename T> __device__ __forceinline__ T rect_d2(T x, T y, T hx, T hy)
{
    T qx = mp_abs(x) - hx, qy = mp_abs(y) - hy;
    T mx = mp_max(qx, T(0)), my = mp_max(qy, T(0));
    return mp_fma(my, my, mx * mx);
}

// One (pose, box) pair.  sep == true: v = squared distance; sep == false: v = minus the
// minimum SAT overlap (<= 0).
// cb, sb = cos / sin of the box's own yaw (1, 0 = axis aligned)
template <typename T> struct BoxRec { T cx, cy, cz, hx, hy, hz, cb, sb; };

// A yaw-rotated static box is handled in the box's own frame: relative position R_b^T (p - c_b) and
// relative rotation theta - theta_b (distances are invariant).  On return (px, py) is relative to the
// box centre, so the pair is evaluated against a box centred at (0, 0, cz).  With cb = 1, sb = 0 every
// value equals the axis-aligned one bit for bit (x*1 and x + (+-0) are exact).
template <typename T>
__device__ __forceinline__ void to_box_frame(const BoxRec<T> &bx, T &px, T &py, T &c, T &s)
{
    const T dxw = px - bx.cx, dyw = py - bx.cy;
    px = mp_fma(bx.sb, dyw, bx.cb * dxw);
    py = mp_fma(bx.cb, dyw, -(bx.sb * dxw));
    const T cr = mp_fma(s, bx.sb, c * bx.cb);
    const T sr = mp_fma(-c, bx.sb, s * bx.cb);
    c = cr; s = sr;
}

template <typename T> struct PairOut {
    T gx, gy, gax, gay, gz, gsat, dx, dy, dz, lx, ly, w1x, w1y, w2x, w2y;
};

template <typename T>
__device__ __forceinline__ void pair_axes(const Pose<T> &p, T ax, T ay, T az, T bcx, T bcy, T bcz, T hbx,
                                           T hby, T hbz, PairOut<T> &o)
{
    T c = p.c, s = p.s, ac = mp_abs(c), as = mp_abs(s);
    o.dx = p.px - bcx; o.dy = p.py - bcy; o.dz = p.pz - bcz;
    o.gz = mp_abs(o.dz) - (az + hbz);
    o.gx = mp_abs(o.dx) - (hbx + p.ex);
    o.gy = mp_abs(o.dy) - (hby + p.ey);
    o.lx = mp_fma(s, o.dy, c * o.dx);
    o.ly = mp_fma(c, o.dy, -(s * o.dx));
    T rbx = mp_fma(as, hby, ac * hbx);
    T rby = mp_fma(ac, hby, as * hbx);
    o.gax = mp_abs(o.lx) - (ax + rbx);
    o.gay = mp_abs(o.ly) - (ay + rby);
    o.gsat = mp_max(mp_max(o.gx, o.gy), mp_max(o.gax, o.gay));
    o.w1x = mp_fma(s, hby, c * hbx);
    o.w1y = mp_fma(c, hby, -(s * hbx));
    o.w2x = mp_fma(-s, hby, c * hbx);
    o.w2y = -mp_fma(c, hby, s * hbx);
}

// record `i` of a box table through the constant address space (scalar loads when i is uniform)
template <typename T> __device__ __forceinline__ BoxRec<T> load_box_uniform(const BoxRec<T> *boxes, int i)
{
    typedef const T __attribute__((address_space(4))) *CPtr;
    CPtr p = (CPtr)(unsigned long long)(boxes + i);
    BoxRec<T> r;
    r.cx = p[0]; r.cy = p[1]; r.cz = p[2]; r.hx = p[3]; r.hy = p[4]; r.hz = p[5];
    r.cb = p[6]; r.sb = p[7];
    return r;
}

// scene value for one pose: min over boxes (boxes pointer and nb are wave-uniform)
template <typename T>
__device__ void tilt_eval(T px, T py, T pz, T c, T s, T ax, T ay, T az, T bcx, T bcy, T bcz, T hbx, T hby, T hbz,
                          const T *__restrict__ R, bool &separated, T &v2, T &pen);

template <typename T>
__device__ __forceinline__ T pose_distance(const Pose<T> &pw, T ax, T ay, T az, const BoxRec<T> *__restrict__ boxes,
                                            int nb, const T *__restrict__ tilt, int &arg_box)
{
    T sep2 = Lim<T>::inf(), pen = Lim<T>::inf();
    int isep = -1, ipen = -1;
    if (nb <= 0) { arg_box = -1; return Lim<T>::inf(); }   // empty scene (and nothing to prefetch)
    // The box table is read-only for the whole launch and the index is wave-uniform: read it
    // through the constant address space so the records arrive over the scalar data path
    // (s_load into SGPRs), and fetch record b+1 while record b is being worked on.
    BoxRec<T> nxt = load_box_uniform(boxes, 0);
    for (int b = 0; b < nb; ++b) {
        const BoxRec<T> bx = nxt;
        nxt = load_box_uniform(boxes, b + 1 < nb ? b + 1 : b);
        if (tilt && bx.sb == T(2)) {   // general rotation
            bool sp; T tv2, tpn;
            tilt_eval(pw.px, pw.py, pw.pz, pw.c, pw.s, ax, ay, az, bx.cx, bx.cy, bx.cz, bx.hx, bx.hy, bx.hz, tilt + 9 * (int)bx.cb, sp, tv2, tpn);
            const bool ts = sp && (tv2 < sep2), tp = !sp && (tpn < pen);
            sep2 = ts ? tv2 : sep2; isep = ts ? b : isep;
            pen = tp ? tpn : pen; ipen = tp ? b : ipen;
            continue;
        }
        T qx = pw.px, qy = pw.py, qc = pw.c, qs = pw.s;
        to_box_frame(bx, qx, qy, qc, qs);
        Pose<T> p;
        pose_setup(p, qx, qy, pw.pz, qc, qs, ax, ay);
        PairOut<T> o;
        pair_axes(p, ax, ay, az, T(0), T(0), bx.cz, bx.hx, bx.hy, bx.hz, o);
        const bool xy_sep = o.gsat > T(0);
        const bool separated = xy_sep || o.gz > T(0);
        T dxy2 = T(0);
        if (__builtin_amdgcn_ballot_w64(xy_sep)) {
            T v = rect_d2(o.dx + p.u1x, o.dy + p.u1y, bx.hx, bx.hy);
            v = mp_min(v, rect_d2(o.dx - p.u1x, o.dy - p.u1y, bx.hx, bx.hy));
            v = mp_min(v, rect_d2(o.dx + p.u2x, o.dy + p.u2y, bx.hx, bx.hy));
            v = mp_min(v, rect_d2(o.dx - p.u2x, o.dy - p.u2y, bx.hx, bx.hy));
            v = mp_min(v, rect_d2(o.lx - o.w1x, o.ly - o.w1y, ax, ay));
            v = mp_min(v, rect_d2(o.lx + o.w1x, o.ly + o.w1y, ax, ay));
            v = mp_min(v, rect_d2(o.lx - o.w2x, o.ly - o.w2y, ax, ay));
            v = mp_min(v, rect_d2(o.lx + o.w2x, o.ly + o.w2y, ax, ay));
            dxy2 = xy_sep ? v : T(0);
        }
        T zs = mp_max(o.gz, T(0));
        T v2 = mp_fma(zs, zs, dxy2);
        T pn = mp_max(o.gsat, o.gz);
        bool bs = separated && (v2 < sep2);
        bool bp = !separated && (pn < pen);
        sep2 = bs ? v2 : sep2;
        isep = bs ? b : isep;
        pen = bp ? pn : pen;
        ipen = bp ? b : ipen;
    }
    if (ipen >= 0) { arg_box = ipen; return pen; }
    arg_box = isep;
    return mp_sqrt(sep2);
}

// ---- culled clearance: conservative fp32 bound + exact evaluation of the survivors ---------
// Ordering key of one exact pair result, so that a single 64-bit unsigned min implements
// "any overlap beats every separation; deepest overlap / smallest squared distance wins":
//   overlapping (value pen <= 0): K = bits(+inf) - bits(|pen|)          (top bit clear)
//   separated   (value v2  >= 0): K = 2^63 | bits(v2)                    (top bit set)
template <typename T> struct Key;
template <> struct Key<double> {
    static constexpr unsigned long long kInfBits = 0x7ff0000000000000ull;
    __device__ static unsigned long long bits(double v) { return (unsigned long long)__double_as_longlong(v); }
    __device__ static double from_bits(unsigned long long b) { return __longlong_as_double((long long)b); }
};
template <> struct Key<float> {
    static constexpr unsigned long long kInfBits = 0x7f800000ull;
    __device__ static unsigned long long bits(float v) { return (unsigned long long)__float_as_uint(v); }
    __device__ static float from_bits(unsigned long long b) { return __uint_as_float((unsigned int)b); }
};
constexpr unsigned long long kSepBit = 0x8000000000000000ull;

template <typename T> __device__ __forceinline__ unsigned long long key_init()
{
    return kSepBit | Key<T>::kInfBits;  // separated by +inf: the value of an empty scene
}
template <typename T> __device__ __forceinline__ T key_to_distance(unsigned long long k)
{
    if (k & kSepBit) return mp_sqrt(Key<T>::from_bits(k & ~kSepBit));
    return -Key<T>::from_bits(Key<T>::kInfBits - k);
}
// squared separation encoded in the key, 0 for an overlap (upper bound used by the culling)
template <typename T> __device__ __forceinline__ T key_to_sep2(unsigned long long k)
{
    return (k & kSepBit) ? Key<T>::from_bits(k & ~kSepBit) : T(0);
}

// exact result of one (pose, box) pair as a key; same arithmetic as the brute-force loop
template <typename T>
__device__ __forceinline__ unsigned long long pair_key(const Pose<T> &p, T ax, T ay, T az, T bcx, T bcy, T bcz, T hbx,
                                                        T hby, T hbz)
{
    PairOut<T> o;
    pair_axes(p, ax, ay, az, bcx, bcy, bcz, hbx, hby, hbz, o);
    const bool xy_sep = o.gsat > T(0);
    const bool separated = xy_sep || o.gz > T(0);
    T v = rect_d2(o.dx + p.u1x, o.dy + p.u1y, hbx, hby);
    v = mp_min(v, rect_d2(o.dx - p.u1x, o.dy - p.u1y, hbx, hby));
    v = mp_min(v, rect_d2(o.dx + p.u2x, o.dy + p.u2y, hbx, hby));
    v = mp_min(v, rect_d2(o.dx - p.u2x, o.dy - p.u2y, hbx, hby));
    v = mp_min(v, rect_d2(o.lx - o.w1x, o.ly - o.w1y, ax, ay));
    v = mp_min(v, rect_d2(o.lx + o.w1x, o.ly + o.w1y, ax, ay));
    v = mp_min(v, rect_d2(o.lx - o.w2x, o.ly - o.w2y, ax, ay));
    v = mp_min(v, rect_d2(o.lx + o.w2x, o.ly + o.w2y, ax, ay));
    T dxy2 = xy_sep ? v : T(0);
    T zs = mp_max(o.gz, T(0));
    T v2 = mp_fma(zs, zs, dxy2);
    T pn = mp_max(o.gsat, o.gz);
    unsigned long long ks = kSepBit | Key<T>::bits(v2);
    unsigned long long kp = Key<T>::kInfBits - Key<T>::bits(mp_abs(pn));
    return separated ? ks : kp;
}

// ---- general static box (any rotation): exact distance against the yaw-rotated mover -----------------------
// A tilted box carries sb = 2 and, in cb, the index of its rotation matrix (world <- box, row major, 9 values) in the
// tilt table.  Overlapping: minus the smallest overlap over the 15 SAT axes; separated: the smallest of 8 + 8
// vertex-to-box and 12 x 12 edge-to-edge squared distances, all in the box's own frame.  Same operation sequence as
// the oracle's tilt_eval.  Deliberately not inlined: it is large, rare (no shipped scenario has a tilted block) and
// only reachable from the ROT = 2 kernel instances.
template <typename T> __device__ __forceinline__ bool box_is_tilted(const BoxRec<T> &bx) { return bx.sb == T(2); }

template <typename T> __device__ __forceinline__ T dot3f(const T *u, const T *v) { return mp_fma(u[2], v[2], mp_fma(u[1], v[1], u[0] * v[0])); }

template <typename T> __device__ __forceinline__ T point_box_d2(const T *v, const T *h)
{
    T q0 = mp_abs(v[0]) - h[0], q1 = mp_abs(v[1]) - h[1], q2 = mp_abs(v[2]) - h[2];
    T m0 = q0 > T(0) ? q0 : T(0), m1 = q1 > T(0) ? q1 : T(0), m2 = q2 > T(0) ? q2 : T(0);
    return mp_fma(m2, m2, mp_fma(m1, m1, m0 * m0));
}

template <typename T> __device__ __forceinline__ T clamp01(T v) { return v < T(0) ? T(0) : (v > T(1) ? T(1) : v); }

template <typename T> __device__ __forceinline__ T seg_seg_d2(const T *P1, const T *d1, const T *P2, const T *d2)
{
    T r[3] = {P1[0] - P2[0], P1[1] - P2[1], P1[2] - P2[2]};
    T a = dot3f(d1, d1), e = dot3f(d2, d2), f = dot3f(d2, r), c = dot3f(d1, r), bb = dot3f(d1, d2);
    T denom = a * e - bb * bb;
    T sp = T(0), tp;
    if (denom > T(0)) sp = clamp01((bb * f - c * e) / denom);
    T tnom = mp_fma(bb, sp, f);
    if (tnom < T(0)) { tp = T(0); sp = clamp01(-c / a); }
    else if (tnom > e) { tp = T(1); sp = clamp01((bb - c) / a); }
    else tp = tnom / e;
    T w[3];
    for (int k = 0; k < 3; ++k) w[k] = mp_fma(-tp, d2[k], mp_fma(sp, d1[k], r[k]));
    return dot3f(w, w);
}

// separated: v2 = squared distance; otherwise pen = largest signed SAT gap (<= 0)
template <typename T>
__device__ __noinline__ void tilt_eval(T px, T py, T pz, T c, T s, T ax, T ay, T az, T bcx, T bcy, T bcz, T hbx, T hby,
                                       T hbz, const T *__restrict__ R, bool &separated, T &v2, T &pen)
{
    const T h[3] = {hbx, hby, hbz}, a[3] = {ax, ay, az};
    const T dw[3] = {px - bcx, py - bcy, pz - bcz};
    T t[3], E[3][3];
    for (int k = 0; k < 3; ++k) {
        t[k] = mp_fma(R[6 + k], dw[2], mp_fma(R[3 + k], dw[1], R[k] * dw[0]));
        E[k][0] = mp_fma(R[3 + k], s, R[k] * c);
        E[k][1] = mp_fma(R[3 + k], c, -(R[k] * s));
        E[k][2] = R[6 + k];
    }
    T e[3][3];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) e[i][k] = E[k][i];
    T gmax = -Lim<T>::inf();
    for (int k = 0; k < 3; ++k) {
        T ra = mp_fma(a[2], mp_abs(E[k][2]), mp_fma(a[1], mp_abs(E[k][1]), a[0] * mp_abs(E[k][0])));
        T g = mp_abs(t[k]) - (h[k] + ra);
        if (g > gmax) gmax = g;
    }
    for (int i = 0; i < 3; ++i) {
        T rb = mp_fma(h[2], mp_abs(E[2][i]), mp_fma(h[1], mp_abs(E[1][i]), h[0] * mp_abs(E[0][i])));
        T g = mp_abs(dot3f(t, e[i])) - (a[i] + rb);
        if (g > gmax) gmax = g;
    }
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
            const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
            T L[3];
            L[k] = T(0); L[k1] = -e[i][k2]; L[k2] = e[i][k1];
            T n2 = dot3f(L, L);
            if (!(n2 > T(1e-12))) continue;
            T ra = mp_fma(a[2], mp_abs(dot3f(e[2], L)), mp_fma(a[1], mp_abs(dot3f(e[1], L)), a[0] * mp_abs(dot3f(e[0], L))));
            T rb = mp_fma(h[2], mp_abs(L[2]), mp_fma(h[1], mp_abs(L[1]), h[0] * mp_abs(L[0])));
            T g = (mp_abs(dot3f(t, L)) - (ra + rb)) / mp_sqrt(n2);
            if (g > gmax) gmax = g;
        }
    if (!(gmax > T(0))) { separated = false; v2 = T(0); pen = gmax; return; }
    T best = Lim<T>::inf();
    for (int m = 0; m < 8; ++m) {
        T sx = (m & 1) ? T(1) : T(-1), sy = (m & 2) ? T(1) : T(-1), sz = (m & 4) ? T(1) : T(-1);
        T v[3];
        for (int k = 0; k < 3; ++k) v[k] = mp_fma(sz * a[2], e[2][k], mp_fma(sy * a[1], e[1][k], mp_fma(sx * a[0], e[0][k], t[k])));
        T d2 = point_box_d2(v, h);
        if (d2 < best) best = d2;
    }
    for (int m = 0; m < 8; ++m) {
        T u[3] = {((m & 1) ? h[0] : -h[0]) - t[0], ((m & 2) ? h[1] : -h[1]) - t[1], ((m & 4) ? h[2] : -h[2]) - t[2]};
        T w[3] = {dot3f(u, e[0]), dot3f(u, e[1]), dot3f(u, e[2])};
        T d2 = point_box_d2(w, a);
        if (d2 < best) best = d2;
    }
    for (int i = 0; i < 3; ++i) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
        T d1[3] = {T(2) * a[i] * e[i][0], T(2) * a[i] * e[i][1], T(2) * a[i] * e[i][2]};
        for (int ma = 0; ma < 4; ++ma) {
            T s1 = (ma & 1) ? a[i1] : -a[i1], s2 = (ma & 2) ? a[i2] : -a[i2];
            T P1[3];
            for (int k = 0; k < 3; ++k) P1[k] = mp_fma(s2, e[i2][k], mp_fma(s1, e[i1][k], mp_fma(-a[i], e[i][k], t[k])));
            for (int k = 0; k < 3; ++k) {
                const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
                T d2v[3] = {T(0), T(0), T(0)};
                d2v[k] = T(2) * h[k];
                for (int mb = 0; mb < 4; ++mb) {
                    T P2[3];
                    P2[k] = -h[k]; P2[k1] = (mb & 1) ? h[k1] : -h[k1]; P2[k2] = (mb & 2) ? h[k2] : -h[k2];
                    T d2 = seg_seg_d2(P1, d1, P2, d2v);
                    if (d2 < best) best = d2;
                }
            }
        }
    }
    separated = true; v2 = best; pen = T(0);
}

template <typename T>
__device__ __forceinline__ unsigned long long tilt_key(T px, T py, T pz, T c, T s, T ax, T ay, T az, const BoxRec<T> &bx,
                                                       const T *__restrict__ tilt)
{
    bool sep; T v2, pen;
    tilt_eval(px, py, pz, c, s, ax, ay, az, bx.cx, bx.cy, bx.cz, bx.hx, bx.hy, bx.hz, tilt + 9 * (int)bx.cb, sep, v2, pen);
    return sep ? (kSepBit | Key<T>::bits(v2)) : (Key<T>::kInfBits - Key<T>::bits(mp_abs(pen)));
}

template <typename T>
__device__ __forceinline__ T tilt_value(T px, T py, T pz, T c, T s, T ax, T ay, T az, const BoxRec<T> &bx, const T *__restrict__ tilt)
{
    bool sep; T v2, pen;
    tilt_eval(px, py, pz, c, s, ax, ay, az, bx.cx, bx.cy, bx.cz, bx.hx, bx.hy, bx.hz, tilt + 9 * (int)bx.cb, sep, v2, pen);
    return sep ? mp_sqrt(v2) : pen;
}

// central differences of the exact pair value in (px, py, pz, theta): the gradient of a tilted pair (see the oracle)
template <typename T> struct TiltFd;
template <> struct TiltFd<double> { static constexpr double h = 1e-6, ch = 0.9999999999995, sh = 9.99999999999833333e-07; };
template <> struct TiltFd<float> { static constexpr float h = 2e-3f, ch = 0.999998000000666666578f, sh = 0.00199999866666693333f; };

template <typename T>
__device__ __forceinline__ void tilt_grad(T px, T py, T pz, T c, T s, T ax, T ay, T az, const BoxRec<T> &bx, const T *__restrict__ tilt,
                                          T &g0, T &g1, T &g2, T &g3)
{
    const T hh = TiltFd<T>::h, inv2h = T(1) / (T(2) * TiltFd<T>::h);
    g0 = (tilt_value(px + hh, py, pz, c, s, ax, ay, az, bx, tilt) - tilt_value(px - hh, py, pz, c, s, ax, ay, az, bx, tilt)) * inv2h;
    g1 = (tilt_value(px, py + hh, pz, c, s, ax, ay, az, bx, tilt) - tilt_value(px, py - hh, pz, c, s, ax, ay, az, bx, tilt)) * inv2h;
    g2 = (tilt_value(px, py, pz + hh, c, s, ax, ay, az, bx, tilt) - tilt_value(px, py, pz - hh, c, s, ax, ay, az, bx, tilt)) * inv2h;
    const T cp = mp_fma(-s, TiltFd<T>::sh, c * TiltFd<T>::ch), sp = mp_fma(c, TiltFd<T>::sh, s * TiltFd<T>::ch);
    const T cm = mp_fma(s, TiltFd<T>::sh, c * TiltFd<T>::ch), sm = mp_fma(-c, TiltFd<T>::sh, s * TiltFd<T>::ch);
    g3 = (tilt_value(px, py, pz, cp, sp, ax, ay, az, bx, tilt) - tilt_value(px, py, pz, cm, sm, ax, ay, az, bx, tilt)) * inv2h;
}

// fp32 record of a box for the bound pass: the box's world AABB as [lo, hi] per axis (rounded outwards);
// pad0 / pad1 carry integers (original index; for group records first / count).  A representative record
// keeps a centre in lox, loy, loz.
struct BoxRec32 { float lox, loy, loz, hix, hiy, hiz, pad0, pad1; };

__device__ __forceinline__ BoxRec32 load_box32_uniform(const BoxRec32 *boxes, int i)
{
    typedef const float __attribute__((address_space(4))) *CPtr;
    CPtr p = (CPtr)(unsigned long long)(boxes + i);
    BoxRec32 r;
    r.lox = p[0]; r.loy = p[1]; r.loz = p[2]; r.hix = p[3]; r.hiy = p[4]; r.hiz = p[5];
    r.pad0 = p[6]; r.pad1 = p[7];
    return r;
}

// Conservative lower bound (fp32) of the squared distance between the mover's world AABB [mlo, mhi] (already
// inflated by `slack`, so that fp32 rounding can never lift the bound above the exactly evaluated value) and a
// box's AABB: per axis max(b.lo - m.hi, m.lo - b.hi, 0), two subtractions and one v_max3_f32.
__device__ __forceinline__ float aabb_lb2(float mlox, float mloy, float mloz, float mhix, float mhiy, float mhiz, const BoxRec32 &b)
{
    const float mx = __builtin_fmaxf(__builtin_fmaxf(b.lox - mhix, mlox - b.hix), 0.f);
    const float my = __builtin_fmaxf(__builtin_fmaxf(b.loy - mhiy, mloy - b.hiy), 0.f);
    const float mz = __builtin_fmaxf(__builtin_fmaxf(b.loz - mhiz, mloz - b.hiz), 0.f);
    return __builtin_fmaf(mz, mz, __builtin_fmaf(my, my, mx * mx));
}

// Two consecutive boxes of the sorted bound table side by side, so that the bound pass tests them with packed fp32
// instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two boxes per instruction, only the max3 and the compare
// stay per box): 17 VALU per pair instead of 13 per box.  idx1 < 0: the pair holds one box only (odd group size).
typedef float v2f __attribute__((ext_vector_type(2)));
struct BoxPair32 { v2f lox, loy, loz, hix, hiy, hiz; int idx0, idx1; float pad0, pad1; };

__device__ __forceinline__ BoxPair32 load_pair_uniform(const BoxPair32 *pairs, int i)
{
    typedef const float __attribute__((address_space(4))) *CPtr;
    CPtr p = (CPtr)(unsigned long long)(pairs + i);
    BoxPair32 r;
    r.lox = v2f{p[0], p[1]}; r.loy = v2f{p[2], p[3]}; r.loz = v2f{p[4], p[5]};
    r.hix = v2f{p[6], p[7]}; r.hiy = v2f{p[8], p[9]}; r.hiz = v2f{p[10], p[11]};
    r.idx0 = __float_as_int(p[12]); r.idx1 = __float_as_int(p[13]);
    r.pad0 = 0.f; r.pad1 = 0.f;
    return r;
}

// aabb_lb2 for both boxes of a pair (same operations per component)
__device__ __forceinline__ v2f aabb_lb2_pair(float mlox, float mloy, float mloz, float mhix, float mhiy, float mhiz, const BoxPair32 &b)
{
    const v2f ax = b.lox - mhix, bx = mlox - b.hix, ay = b.loy - mhiy, by = mloy - b.hiy, az = b.loz - mhiz, bz = mloz - b.hiz;
    const v2f mx = v2f{__builtin_fmaxf(__builtin_fmaxf(ax.x, bx.x), 0.f), __builtin_fmaxf(__builtin_fmaxf(ax.y, bx.y), 0.f)};
    const v2f my = v2f{__builtin_fmaxf(__builtin_fmaxf(ay.x, by.x), 0.f), __builtin_fmaxf(__builtin_fmaxf(ay.y, by.y), 0.f)};
    const v2f mz = v2f{__builtin_fmaxf(__builtin_fmaxf(az.x, bz.x), 0.f), __builtin_fmaxf(__builtin_fmaxf(az.y, bz.y), 0.f)};
    return __builtin_elementwise_fma(mz, mz, __builtin_elementwise_fma(my, my, mx * mx));
}

// d(pair)/d(px,py,pz,theta[rad]) for the deciding box (mirrors the oracle's pair_grad)
template <typename T>
__device__ __forceinline__ void pair_grad_local(const Pose<T> &p, T ax, T ay, T az, T bcx, T bcy, const BoxRec<T> bx,
                                                 T &g0, T &g1, T &g2, T &g3);

// world pose (px,py,pz,c,s) against one box; `rot` (wave-uniform) = the scene holds yaw-rotated boxes
template <typename T>
__device__ __forceinline__ void pair_grad(T px, T py, T pz, T c, T s, T ax, T ay, T az, const BoxRec<T> bx, bool rot,
                                           T &g0, T &g1, T &g2, T &g3)
{
    Pose<T> p;
    if (rot) {
        to_box_frame(bx, px, py, c, s);
        pose_setup(p, px, py, pz, c, s, ax, ay);
        T l0, l1;
        pair_grad_local(p, ax, ay, az, T(0), T(0), bx, l0, l1, g2, g3);
        g0 = bx.cb * l0 - bx.sb * l1;   // box frame -> world
        g1 = bx.sb * l0 + bx.cb * l1;
    } else {
        pose_setup(p, px, py, pz, c, s, ax, ay);
        pair_grad_local(p, ax, ay, az, bx.cx, bx.cy, bx, g0, g1, g2, g3);
    }
}

template <typename T>
__device__ __forceinline__ void pair_grad_local(const Pose<T> &p, T ax, T ay, T az, T bcx, T bcy, const BoxRec<T> bx,
                                                 T &g0, T &g1, T &g2, T &g3)
{
    PairOut<T> o;
    pair_axes(p, ax, ay, az, bcx, bcy, bx.cz, bx.hx, bx.hy, bx.hz, o);
    const T c = p.c, s = p.s;
    g0 = g1 = g2 = g3 = T(0);
    const bool xy_sep = o.gsat > T(0);
    if (xy_sep || o.gz > T(0)) {
        T v[8];
        v[0] = rect_d2(o.dx + p.u1x, o.dy + p.u1y, bx.hx, bx.hy);
        v[1] = rect_d2(o.dx - p.u1x, o.dy - p.u1y, bx.hx, bx.hy);
        v[2] = rect_d2(o.dx + p.u2x, o.dy + p.u2y, bx.hx, bx.hy);
        v[3] = rect_d2(o.dx - p.u2x, o.dy - p.u2y, bx.hx, bx.hy);
        v[4] = rect_d2(o.lx - o.w1x, o.ly - o.w1y, ax, ay);
        v[5] = rect_d2(o.lx + o.w1x, o.ly + o.w1y, ax, ay);
        v[6] = rect_d2(o.lx - o.w2x, o.ly - o.w2y, ax, ay);
        v[7] = rect_d2(o.lx + o.w2x, o.ly + o.w2y, ax, ay);
        T dxy2 = v[0];
        int f = 0;
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            bool lt = v[i] < dxy2;
            dxy2 = lt ? v[i] : dxy2;
            f = lt ? i : f;
        }
        if (!xy_sep) dxy2 = T(0);
        T zs = mp_max(o.gz, T(0));
        const T d2 = mp_fma(zs, zs, dxy2);
        if (!(d2 > T(0))) return;
        T d, inv_d;
        sqrt_rsqrt(d2, d, inv_d);
        if (o.gz > T(0)) g2 = sgn(o.dz) * o.gz * inv_d;
        if (!xy_sep) return;
        if (f < 4) {
            T rx = (f < 2) ? p.u1x : p.u2x, ry = (f < 2) ? p.u1y : p.u2y;
            if (f & 1) { rx = -rx; ry = -ry; }
            T X = o.dx + rx, Y = o.dy + ry;
            T qx = mp_abs(X) - bx.hx, qy = mp_abs(Y) - bx.hy;
            T mx = qx > T(0) ? sgn(X) * qx : T(0);
            T my = qy > T(0) ? sgn(Y) * qy : T(0);
            g0 = mx * inv_d;
            g1 = my * inv_d;
            g3 = (my * rx - mx * ry) * inv_d;
        } else {
            T wx = (f < 6) ? o.w1x : o.w2x, wy = (f < 6) ? o.w1y : o.w2y;
            if (!(f & 1)) { wx = -wx; wy = -wy; }
            T tx = o.lx + wx, ty = o.ly + wy;
            T qx = mp_abs(tx) - ax, qy = mp_abs(ty) - ay;
            T mx = qx > T(0) ? sgn(tx) * qx : T(0);
            T my = qy > T(0) ? sgn(ty) * qy : T(0);
            g0 = (c * mx - s * my) * inv_d;
            g1 = (s * mx + c * my) * inv_d;
            g3 = (mx * ty - my * tx) * inv_d;
        }
    } else {
        int a = 0;
        T best = o.gx;
        if (o.gy > best) { best = o.gy; a = 1; }
        if (o.gax > best) { best = o.gax; a = 2; }
        if (o.gay > best) { best = o.gay; a = 3; }
        if (o.gz > best) { best = o.gz; a = 4; }
        T sc = sgn(c), ss = sgn(s);
        if (a == 0) {
            g0 = sgn(o.dx);
            g3 = -(-sc * s * ax + ss * c * ay);
        } else if (a == 1) {
            g1 = sgn(o.dy);
            g3 = -(ss * c * ax - sc * s * ay);
        } else if (a == 2) {
            T sl = sgn(o.lx);
            g0 = sl * c; g1 = sl * s;
            g3 = sl * o.ly - (-sc * s * bx.hx + ss * c * bx.hy);
        } else if (a == 3) {
            T sl = sgn(o.ly);
            g0 = -sl * s; g1 = sl * c;
            g3 = -sl * o.lx - (ss * c * bx.hx - sc * s * bx.hy);
        } else {
            g2 = sgn(o.dz);
        }
    }
}

// ---- point model -----------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T point_distance(T px, T py, T pz, const BoxRec<T> *__restrict__ boxes, int nb, T radius,
                                             bool rot, const T *__restrict__ tilt, int &arg_box)
{
    T out2 = Lim<T>::inf(), in = Lim<T>::inf();
    int iout = -1, iin = -1;
    if (nb <= 0) { arg_box = -1; return Lim<T>::inf(); }   // empty scene (and nothing to prefetch)
    // boxes and b are wave-uniform: records come over the scalar data path, one record ahead (as in pose_distance)
    BoxRec<T> nxt = load_box_uniform(boxes, 0);
    for (int b = 0; b < nb; ++b) {
        const BoxRec<T> bx = nxt;
        nxt = load_box_uniform(boxes, b + 1 < nb ? b + 1 : b);
        T ex = px - bx.cx, ey = py - bx.cy, ez = pz - bx.cz;
        if (rot) {
            const T exw = ex, eyw = ey, ezw = ez;
            if (tilt && bx.sb == T(2)) {   // e = R^T (p - c)
                const T *R = tilt + 9 * (int)bx.cb;
                ex = mp_fma(R[6], ezw, mp_fma(R[3], eyw, R[0] * exw));
                ey = mp_fma(R[7], ezw, mp_fma(R[4], eyw, R[1] * exw));
                ez = mp_fma(R[8], ezw, mp_fma(R[5], eyw, R[2] * exw));
            } else {
                ex = mp_fma(bx.sb, eyw, bx.cb * exw);
                ey = mp_fma(bx.cb, eyw, -(bx.sb * exw));
            }
        }
        T qx = mp_abs(ex) - bx.hx, qy = mp_abs(ey) - bx.hy, qz = mp_abs(ez) - bx.hz;
        bool outside = qx > T(0) || qy > T(0) || qz > T(0);
        T mx = mp_max(qx, T(0)), my = mp_max(qy, T(0)), mz = mp_max(qz, T(0));
        T v = mp_fma(mz, mz, mp_fma(my, my, mx * mx));
        T m = mp_max(mp_max(qx, qy), qz);
        bool bo = outside && v < out2;
        bool bi = !outside && m < in;
        out2 = bo ? v : out2;
        iout = bo ? b : iout;
        in = bi ? m : in;
        iin = bi ? b : iin;
    }
    if (iin >= 0) { arg_box = iin; return in; }
    arg_box = iout;
    return mp_sqrt(out2) - radius;
}

template <typename T>
__device__ __forceinline__ void point_grad(T px, T py, T pz, const BoxRec<T> bx, bool rot, const T *__restrict__ tilt,
                                            T &g0, T &g1, T &g2)
{
    T ex = px - bx.cx, ey = py - bx.cy, ez = pz - bx.cz;
    const bool tilted = rot && tilt && bx.sb == T(2);
    if (tilted) {
        const T *R = tilt + 9 * (int)bx.cb;
        const T exw = ex, eyw = ey, ezw = ez;
        ex = mp_fma(R[6], ezw, mp_fma(R[3], eyw, R[0] * exw));
        ey = mp_fma(R[7], ezw, mp_fma(R[4], eyw, R[1] * exw));
        ez = mp_fma(R[8], ezw, mp_fma(R[5], eyw, R[2] * exw));
    } else if (rot) {
        const T exw = ex, eyw = ey;
        ex = mp_fma(bx.sb, eyw, bx.cb * exw);
        ey = mp_fma(bx.cb, eyw, -(bx.sb * exw));
    }
    T qx = mp_abs(ex) - bx.hx, qy = mp_abs(ey) - bx.hy, qz = mp_abs(ez) - bx.hz;
    g0 = g1 = g2 = T(0);
    if (qx > T(0) || qy > T(0) || qz > T(0)) {
        T mx = mp_max(qx, T(0)), my = mp_max(qy, T(0)), mz = mp_max(qz, T(0));
        const T d2 = mp_fma(mz, mz, mp_fma(my, my, mx * mx));
        if (!(d2 > T(0))) return;
        T d, inv_d;
        sqrt_rsqrt(d2, d, inv_d);
        g0 = sgn(ex) * mx * inv_d; g1 = sgn(ey) * my * inv_d; g2 = sgn(ez) * mz * inv_d;
    } else {
        int a = 0;
        T best = qx;
        if (qy > best) { best = qy; a = 1; }
        if (qz > best) { best = qz; a = 2; }
        T sg = sgn(a == 0 ? ex : (a == 1 ? ey : ez));
        g0 = a == 0 ? sg : T(0); g1 = a == 1 ? sg : T(0); g2 = a == 2 ? sg : T(0);
    }
    if (tilted) {   // box frame -> world: g_w = R g_l
        const T *R = tilt + 9 * (int)bx.cb;
        const T l0 = g0, l1 = g1, l2 = g2;
        g0 = mp_fma(R[2], l2, mp_fma(R[1], l1, R[0] * l0));
        g1 = mp_fma(R[5], l2, mp_fma(R[4], l1, R[3] * l0));
        g2 = mp_fma(R[8], l2, mp_fma(R[7], l1, R[6] * l0));
    } else if (rot) {   // box frame -> world
        const T l0 = g0, l1 = g1;
        g0 = bx.cb * l0 - bx.sb * l1;
        g1 = bx.sb * l0 + bx.cb * l1;
    }
}

// ---- wave helpers ----------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync()
{
    // the 64 lanes of one wave exchange data through their private LDS slice: LDS
    // operations of a wave complete in issue order, this only pins the compiler's order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cross-lane moves through the DPP network (no LDS round trip).  A 64-bit value moves as two dwords.
template <int CTRL, int ROW_MASK, bool ZERO_FILL> __device__ __forceinline__ int dpp32(int v)
{
    return __builtin_amdgcn_update_dpp(ZERO_FILL ? 0 : v, v, CTRL, ROW_MASK, 0xf, ZERO_FILL);
}
template <int CTRL, int ROW_MASK, bool ZERO_FILL> __device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(dpp32<CTRL, ROW_MASK, ZERO_FILL>(__float_as_int(v)));
}
template <int CTRL, int ROW_MASK, bool ZERO_FILL> __device__ __forceinline__ double dpp_mov(double v)
{
    int lo = dpp32<CTRL, ROW_MASK, ZERO_FILL>(__double2loint(v));
    int hi = dpp32<CTRL, ROW_MASK, ZERO_FILL>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK, bool ZERO_FILL> __device__ __forceinline__ int dpp_mov(int v)
{
    return dpp32<CTRL, ROW_MASK, ZERO_FILL>(v);
}
__device__ __forceinline__ int read_lane63(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ float read_lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__device__ __forceinline__ double read_lane63(double v)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Sum over the 64 lanes, result wave-uniform.  quad swaps -> half-row mirror -> row mirror give every
// lane its row total; row_bcast:15 / row_bcast:31 carry the totals up to lane 63.
template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
    v += dpp_mov<0xB1, 0xf, true>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf, true>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf, true>(v);  // row_half_mirror
    v += dpp_mov<0x140, 0xf, true>(v);  // row_mirror
    v += dpp_mov<0x142, 0xa, true>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_mov<0x143, 0xc, true>(v);  // row_bcast:31 into rows 2 and 3
    return read_lane63(v);
}
template <typename T> __device__ __forceinline__ T lane_min(T a, T b) { return b < a ? b : a; }
__device__ __forceinline__ int lane_min_fast(int a, int b) { return b < a ? b : a; }
// floating-point operands are never NaN here (a running minimum only ever takes a value that compared smaller, and a CEM
// cost has its NaNs replaced by +inf before the reduction): one v_min instead of a compare and two selects
__device__ __forceinline__ float lane_min_fast(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double lane_min_fast(double a, double b) { return __builtin_fmin(a, b); }
// cross-lane move whose every lane has a source (quad permutations, row mirrors): no `old` value to preserve, so no
// register copy in front of the DPP move
template <int CTRL> __device__ __forceinline__ int dpp_full(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ float dpp_full(float v) { return __int_as_float(dpp_full<CTRL>(__float_as_int(v))); }
template <int CTRL> __device__ __forceinline__ double dpp_full(double v)
{
    return __hiloint2double(dpp_full<CTRL>(__double2hiint(v)), dpp_full<CTRL>(__double2loint(v)));
}
template <typename T> __device__ __forceinline__ T wave_min(T v)
{
    v = lane_min_fast(v, dpp_full<0xB1>(v));    // quad_perm [1,0,3,2]
    v = lane_min_fast(v, dpp_full<0x4E>(v));    // quad_perm [2,3,0,1]
    v = lane_min_fast(v, dpp_full<0x141>(v));   // row_half_mirror
    v = lane_min_fast(v, dpp_full<0x140>(v));   // row_mirror
    v = lane_min_fast(v, dpp_mov<0x142, 0xa, false>(v));   // row_bcast:15 into rows 1 and 3 (the other rows keep their value)
    v = lane_min_fast(v, dpp_mov<0x143, 0xc, false>(v));   // row_bcast:31 into rows 2 and 3
    return read_lane63(v);
}

}  // namespace mp
