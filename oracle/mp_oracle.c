/*
 * mp_oracle.c — CPU restatement (plain scalar C) of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (motion_planning_amd/, the HIP
 * library) may include, link or call this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker / the CPU baseline.
 *
 * What it restates (paths relative to the reference checkout):
 *   objective_single          geom/spline_opt.py:500-641   (15-term objective)
 *   3-term objective_single   geom/spline_opt.py:797-819
 *   path_length               geom/spline_opt.py:54-60
 *   curvature_cost            geom/spline_opt.py:63-79     (two np.gradient passes)
 *   yaw_smoothness_cost       geom/spline_opt.py:100-106
 *   _goal_approach_alignment_cost geom/spline_opt.py:117-145
 *   yaw_deg_to_quat           geom/spline_opt.py:37-40  + quat_to_rot geom/utils.py:3-14
 *   Scene.signed_distance     geom/scene.py:50-68   (point model)
 *   Scene.signed_distance_block geom/scene.py:70-122 (box model)
 *
 * Parity status.  Everything downstream of the clearance profile d_safe, and the
 * spline sampling, is pinned against the real reference module (tests/golden/, made
 * by tools/gen_golden.py which loads /root/reference/geom/spline_opt.py).  The
 * DISTANCE VALUES themselves are "parity unpinned": the reference delegates them to
 * python-fcl (un-pinned version, not installable here, no fixtures in the reference),
 * so this file DEFINES them as exact analytic geometry:
 *   box model  : yaw-rotated box vs axis-aligned box.  Separated: exact Euclidean
 *                distance (2-D rectangle-rectangle distance by the 8 vertex-to-
 *                rectangle tests, combined with the z gap).  Overlapping/touching:
 *                minus the minimum SAT overlap over the 5 unique axes (0 when
 *                touching).  Scene value = min over boxes; empty scene = +inf.
 *   point model: exact box SDF; outside values minus the 1e-6 probe radius.
 * The analytic gradient has no reference counterpart (the reference is derivative-
 * free); it is checked against central differences of this file's own cost.
 *
 * Arithmetic contract shared with the HIP kernels (so that flags are bit-exact):
 * IEEE + - * / sqrt, explicit fma() exactly where written below, no other
 * contraction (compile with -ffp-contract=off), and the sin/cos below instead of
 * libm's.  The sample matrix M is an input (never rebuilt here).
 *
 * Build: oracle/Makefile (compiled twice: double, and -DORC_F32 float).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mp_api.h"

#ifdef ORC_F32
typedef float real;
#define RL(x) x##f
#define FMA fmaf
#define SQRT sqrtf
#define FABS fabsf
#define RINT rintf
#define ORC_NAME(n) n##_f32
#else
typedef double real;
#define RL(x) x
#define FMA fma
#define SQRT sqrt
#define FABS fabs
#define RINT rint
#define ORC_NAME(n) n##_f64
#endif

#define INF_R ((real)INFINITY)

/* ------------------------------------------------------------------------------
 * sin/cos of the half yaw angle.  Cody-Waite reduction by pi/2 in two fma steps and
 * the classic degree-13/12 minimax kernels on [-pi/4, pi/4] (coefficients as
 * published in fdlibm's k_sin.c / k_cos.c).  Valid for |angle| up to ~1e5 rad.
 * ---------------------------------------------------------------------------- */
static void orc_sincos(real h, real *sn, real *cs)
{
    const real TWO_OVER_PI = RL(0.63661977236758134308);
#ifdef ORC_F32
    const real PIO2_HI = 1.57079637050628662109e+00f;
    const real PIO2_LO = -4.37113882867379288655e-08f;
#else
    const real PIO2_HI = 1.57079632679489655800e+00;
    const real PIO2_LO = 6.12323399573676603587e-17;
#endif
    const real S1 = RL(-1.66666666666666324348e-01), S2 = RL(8.33333333332248946124e-03),
               S3 = RL(-1.98412698298579493134e-04), S4 = RL(2.75573137070700676789e-06),
               S5 = RL(-2.50507602534068634195e-08), S6 = RL(1.58969099521155010221e-10);
    const real C1 = RL(4.16666666666666019037e-02), C2 = RL(-1.38888888888741095749e-03),
               C3 = RL(2.48015872894767294178e-05), C4 = RL(-2.75573143513906633035e-07),
               C5 = RL(2.08757232129817482790e-09), C6 = RL(-1.13596475577881948265e-11);
    real k = RINT(h * TWO_OVER_PI);
    real r = FMA(-k, PIO2_HI, h);
    r = FMA(-k, PIO2_LO, r);
    int q = (int)((long long)k) & 3;
    real z = r * r;
    real ps = FMA(z, S6, S5);
    ps = FMA(z, ps, S4);
    ps = FMA(z, ps, S3);
    ps = FMA(z, ps, S2);
    ps = FMA(z, ps, S1);
    real sr = FMA(r * z, ps, r);
    real pc = FMA(z, C6, C5);
    pc = FMA(z, pc, C4);
    pc = FMA(z, pc, C3);
    pc = FMA(z, pc, C2);
    pc = FMA(z, pc, C1);
    real cr = FMA(z * z, pc, FMA(RL(-0.5), z, RL(1.0)));
    switch (q) {
    case 0: *sn = sr; *cs = cr; break;
    case 1: *sn = cr; *cs = -sr; break;
    case 2: *sn = -sr; *cs = -cr; break;
    default: *sn = -cr; *cs = sr; break;
    }
}

/* yaw (degrees) -> (cos yaw, sin yaw) composed exactly like the reference composes
 * its rotation matrix: quaternion (0,0,sin h,cos h) with h = 0.5*deg2rad(yaw)
 * (spline_opt.py:37-40), then R00 = 1 - 2 z^2, R10 = 2 w z (utils.py:3-14). */
static void orc_yaw_rot(real yaw_deg, real *c, real *s)
{
    const real DEG2RAD = RL(0.017453292519943295);
    real h = RL(0.5) * (yaw_deg * DEG2RAD);
    real zq, wq;
    orc_sincos(h, &zq, &wq);
    *c = RL(1.0) - RL(2.0) * (zq * zq);
    *s = RL(2.0) * (wq * zq);
}

static real sgn(real v) { return v < RL(0.0) ? RL(-1.0) : RL(1.0); }
/* np.maximum(0, v): propagates NaN */
static real npmax0(real v) { return !(v <= RL(0.0)) ? v : RL(0.0); }
static real maxr(real a, real b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------
 * Box model: one moving-box pose against the boxes of a scene.
 * ---------------------------------------------------------------------------- */
typedef struct {
    real px, py, pz, c, s;
    real ex, ey;              /* xy AABB half extents of the rotated rectangle      */
    real u1x, u1y, u2x, u2y;  /* rotated corner offsets (+ax,+ay), (+ax,-ay)        */
    real ax, ay, az;
} pose_t;

static void pose_setup_cs(pose_t *p, real px, real py, real pz, real c, real s, const real a[3])
{
    p->px = px; p->py = py; p->pz = pz;
    p->ax = a[0]; p->ay = a[1]; p->az = a[2];
    p->c = c; p->s = s;
    real ac = FABS(c), as = FABS(s);
    p->ex = FMA(as, p->ay, ac * p->ax);
    p->ey = FMA(ac, p->ay, as * p->ax);
    p->u1x = FMA(-s, p->ay, c * p->ax);
    p->u1y = FMA(c, p->ay, s * p->ax);
    p->u2x = FMA(s, p->ay, c * p->ax);
    p->u2y = FMA(-c, p->ay, s * p->ax);
}

static void pose_setup(pose_t *p, real px, real py, real pz, real yaw_deg, const real a[3])
{
    real c, s;
    orc_yaw_rot(yaw_deg, &c, &s);
    pose_setup_cs(p, px, py, pz, c, s, a);
}

/* squared distance from point (x,y) to the origin-centred rectangle (hx,hy) */
static real rect_d2(real x, real y, real hx, real hy)
{
    real qx = FABS(x) - hx, qy = FABS(y) - hy;
    real mx = qx > RL(0.0) ? qx : RL(0.0), my = qy > RL(0.0) ? qy : RL(0.0);
    return FMA(my, my, mx * mx);
}

typedef struct {
    real gx, gy, gax, gay, gz, gsat;
    real dx, dy, dz, lx, ly;
    real w1x, w1y, w2x, w2y;
    real v2;      /* separated: squared distance; overlapping: unused               */
    real pen;     /* overlapping: max(gsat, gz) <= 0                                 */
    int separated;
    int feature;  /* separated && gsat > 0: argmin vertex test 0..7                 */
    pose_t q;     /* the pose expressed in the box's own (yaw-rotated) frame        */
} pair_t;

/* Box record: cx,cy,cz, hx,hy,hz, cb,sb  (cb,sb = cos/sin of the box's own yaw; 1,0 = axis aligned).
 * A yaw-rotated static box is handled by expressing the pair in the box frame: relative position
 * R_b^T (p - c_b) and relative rotation theta_a - theta_b; distances are invariant.  For cb = 1,
 * sb = 0 every expression below reduces bit-for-bit to the axis-aligned formulas. */
/* A box tilted out of the horizontal plane (general unit quaternion) carries sb = 2 and its rotation matrix
 * R (world <- box, row major) in b[9..17]; it goes through tilt_eval below. */
#define BOX_STRIDE 18
#define BOX_IS_TILTED(b) ((b)[7] == RL(2.0))

static void tilt_eval(const pose_t *pw, const real *b, pair_t *o);

static void pair_eval(const pose_t *pw, const real *b, pair_t *o)
{
    if (BOX_IS_TILTED(b)) { tilt_eval(pw, b, o); return; }
    const real cb = b[6], sb = b[7];
    const real a3[3] = {pw->ax, pw->ay, pw->az};
    real dxw = pw->px - b[0], dyw = pw->py - b[1];
    real dx = FMA(sb, dyw, cb * dxw);
    real dy = FMA(cb, dyw, -(sb * dxw));
    real cr = FMA(pw->s, sb, pw->c * cb);
    real sr = FMA(-pw->c, sb, pw->s * cb);
    pose_setup_cs(&o->q, dx, dy, pw->pz - b[2], cr, sr, a3);
    const pose_t *p = &o->q;
    real c = p->c, s = p->s, ac = FABS(c), as = FABS(s);
    real hbx = b[3], hby = b[4], hbz = b[5];
    real dz = p->pz;
    o->dx = dx; o->dy = dy; o->dz = dz;
    o->gz = FABS(dz) - (p->az + hbz);
    o->gx = FABS(dx) - (hbx + p->ex);
    o->gy = FABS(dy) - (hby + p->ey);
    real lx = FMA(s, dy, c * dx);
    real ly = FMA(c, dy, -(s * dx));
    o->lx = lx; o->ly = ly;
    real rbx = FMA(as, hby, ac * hbx);
    real rby = FMA(ac, hby, as * hbx);
    o->gax = FABS(lx) - (p->ax + rbx);
    o->gay = FABS(ly) - (p->ay + rby);
    o->gsat = maxr(maxr(o->gx, o->gy), maxr(o->gax, o->gay));
    o->w1x = FMA(s, hby, c * hbx);
    o->w1y = FMA(c, hby, -(s * hbx));
    o->w2x = FMA(-s, hby, c * hbx);
    o->w2y = -FMA(c, hby, s * hbx);
    o->feature = -1;
    if (o->gsat > RL(0.0) || o->gz > RL(0.0)) {
        real dxy2 = RL(0.0);
        if (o->gsat > RL(0.0)) {
            real v[8];
            v[0] = rect_d2(dx + p->u1x, dy + p->u1y, hbx, hby);
            v[1] = rect_d2(dx - p->u1x, dy - p->u1y, hbx, hby);
            v[2] = rect_d2(dx + p->u2x, dy + p->u2y, hbx, hby);
            v[3] = rect_d2(dx - p->u2x, dy - p->u2y, hbx, hby);
            v[4] = rect_d2(lx - o->w1x, ly - o->w1y, p->ax, p->ay);
            v[5] = rect_d2(lx + o->w1x, ly + o->w1y, p->ax, p->ay);
            v[6] = rect_d2(lx - o->w2x, ly - o->w2y, p->ax, p->ay);
            v[7] = rect_d2(lx + o->w2x, ly + o->w2y, p->ax, p->ay);
            dxy2 = v[0]; o->feature = 0;
            for (int f = 1; f < 8; ++f)
                if (v[f] < dxy2) { dxy2 = v[f]; o->feature = f; }
        }
        real zs = o->gz > RL(0.0) ? o->gz : RL(0.0);
        o->v2 = FMA(zs, zs, dxy2);
        o->separated = 1;
        o->pen = RL(0.0);
    } else {
        o->separated = 0;
        o->v2 = RL(0.0);
        o->pen = maxr(o->gsat, o->gz);
    }
}


/* ------------------------------------------------------------------------------
 * General static box (any rotation) against the yaw-rotated mover — exact geometry:
 *   overlapping: minus the smallest overlap over the 15 SAT axes (3 + 3 face normals, 9 edge cross products);
 *   separated:   the smallest of 8 + 8 vertex-to-box and 12 x 12 edge-to-edge squared distances.
 * For two boxes that share the z axis this is the same value as the yaw-only path (the extra axes and
 * features coincide with the ones used there).  Everything is evaluated in the box's own frame.
 * ---------------------------------------------------------------------------- */
static real dot3f(const real *u, const real *v) { return FMA(u[2], v[2], FMA(u[1], v[1], u[0] * v[0])); }

static real point_box_d2(const real *v, const real *h)
{
    real q0 = FABS(v[0]) - h[0], q1 = FABS(v[1]) - h[1], q2 = FABS(v[2]) - h[2];
    real m0 = q0 > RL(0.0) ? q0 : RL(0.0), m1 = q1 > RL(0.0) ? q1 : RL(0.0), m2 = q2 > RL(0.0) ? q2 : RL(0.0);
    return FMA(m2, m2, FMA(m1, m1, m0 * m0));
}

static real clamp01(real v) { return v < RL(0.0) ? RL(0.0) : (v > RL(1.0) ? RL(1.0) : v); }

/* squared distance between segments P1 + s d1 and P2 + t d2, s, t in [0, 1] (both of non-zero length) */
static real seg_seg_d2(const real *P1, const real *d1, const real *P2, const real *d2)
{
    real r[3] = {P1[0] - P2[0], P1[1] - P2[1], P1[2] - P2[2]};
    real a = dot3f(d1, d1), e = dot3f(d2, d2), f = dot3f(d2, r), c = dot3f(d1, r), bb = dot3f(d1, d2);
    real denom = a * e - bb * bb;
    real sp = RL(0.0), tp;
    if (denom > RL(0.0)) sp = clamp01((bb * f - c * e) / denom);
    real tnom = FMA(bb, sp, f);
    if (tnom < RL(0.0)) { tp = RL(0.0); sp = clamp01(-c / a); }
    else if (tnom > e) { tp = RL(1.0); sp = clamp01((bb - c) / a); }
    else tp = tnom / e;
    real w[3];
    for (int k = 0; k < 3; ++k) w[k] = FMA(-tp, d2[k], FMA(sp, d1[k], r[k]));
    return dot3f(w, w);
}

static void tilt_eval(const pose_t *pw, const real *b, pair_t *o)
{
    const real *R = b + 9;
    const real h[3] = {b[3], b[4], b[5]}, a[3] = {pw->ax, pw->ay, pw->az};
    const real dw[3] = {pw->px - b[0], pw->py - b[1], pw->pz - b[2]};
    real t[3], E[3][3]; /* E[k][i]: component k (box frame) of the mover's axis i */
    for (int k = 0; k < 3; ++k) {
        t[k] = FMA(R[6 + k], dw[2], FMA(R[3 + k], dw[1], R[k] * dw[0]));
        E[k][0] = FMA(R[3 + k], pw->s, R[k] * pw->c);
        E[k][1] = FMA(R[3 + k], pw->c, -(R[k] * pw->s));
        E[k][2] = R[6 + k];
    }
    real e[3][3]; /* e[i] = mover axis i as a vector in the box frame */
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) e[i][k] = E[k][i];
    /* --- 15-axis SAT: largest signed gap --- */
    real gmax = -INF_R;
    for (int k = 0; k < 3; ++k) {
        real ra = FMA(a[2], FABS(E[k][2]), FMA(a[1], FABS(E[k][1]), a[0] * FABS(E[k][0])));
        real g = FABS(t[k]) - (h[k] + ra);
        if (g > gmax) gmax = g;
    }
    for (int i = 0; i < 3; ++i) {
        real rb = FMA(h[2], FABS(E[2][i]), FMA(h[1], FABS(E[1][i]), h[0] * FABS(E[0][i])));
        real g = FABS(dot3f(t, e[i])) - (a[i] + rb);
        if (g > gmax) gmax = g;
    }
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
            /* L = e_i x u_k (u_k = unit vector k of the box frame) */
            const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
            real L[3];
            L[k] = RL(0.0); L[k1] = -e[i][k2]; L[k2] = e[i][k1];
            real n2 = dot3f(L, L);
            if (!(n2 > RL(1e-12))) continue; /* parallel edges: the face normals decide */
            real ra = FMA(a[2], FABS(dot3f(e[2], L)), FMA(a[1], FABS(dot3f(e[1], L)), a[0] * FABS(dot3f(e[0], L))));
            real rb = FMA(h[2], FABS(L[2]), FMA(h[1], FABS(L[1]), h[0] * FABS(L[0])));
            real g = (FABS(dot3f(t, L)) - (ra + rb)) / SQRT(n2);
            if (g > gmax) gmax = g;
        }
    o->feature = -1;
    o->gsat = gmax; o->gz = gmax;
    if (!(gmax > RL(0.0))) {
        o->separated = 0; o->v2 = RL(0.0); o->pen = gmax;
        return;
    }
    /* --- separated: closest features --- */
    real best = INF_R;
    for (int m = 0; m < 8; ++m) { /* vertices of the mover against the box */
        real sx = (m & 1) ? RL(1.0) : RL(-1.0), sy = (m & 2) ? RL(1.0) : RL(-1.0), sz = (m & 4) ? RL(1.0) : RL(-1.0);
        real v[3];
        for (int k = 0; k < 3; ++k) v[k] = FMA(sz * a[2], e[2][k], FMA(sy * a[1], e[1][k], FMA(sx * a[0], e[0][k], t[k])));
        real d2 = point_box_d2(v, h);
        if (d2 < best) best = d2;
    }
    for (int m = 0; m < 8; ++m) { /* vertices of the box against the mover (in the mover's frame) */
        real u[3] = {((m & 1) ? h[0] : -h[0]) - t[0], ((m & 2) ? h[1] : -h[1]) - t[1], ((m & 4) ? h[2] : -h[2]) - t[2]};
        real w[3] = {dot3f(u, e[0]), dot3f(u, e[1]), dot3f(u, e[2])};
        real d2 = point_box_d2(w, a);
        if (d2 < best) best = d2;
    }
    for (int i = 0; i < 3; ++i) { /* edges of the mover along its axis i ... */
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
        real d1[3] = {RL(2.0) * a[i] * e[i][0], RL(2.0) * a[i] * e[i][1], RL(2.0) * a[i] * e[i][2]};
        for (int ma = 0; ma < 4; ++ma) {
            real s1 = (ma & 1) ? a[i1] : -a[i1], s2 = (ma & 2) ? a[i2] : -a[i2];
            real P1[3];
            for (int k = 0; k < 3; ++k) P1[k] = FMA(s2, e[i2][k], FMA(s1, e[i1][k], FMA(-a[i], e[i][k], t[k])));
            for (int k = 0; k < 3; ++k) { /* ... against the edges of the box along its axis k */
                const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
                real d2v[3] = {RL(0.0), RL(0.0), RL(0.0)};
                d2v[k] = RL(2.0) * h[k];
                for (int mb = 0; mb < 4; ++mb) {
                    real P2[3];
                    P2[k] = -h[k]; P2[k1] = (mb & 1) ? h[k1] : -h[k1]; P2[k2] = (mb & 2) ? h[k2] : -h[k2];
                    real d2 = seg_seg_d2(P1, d1, P2, d2v);
                    if (d2 < best) best = d2;
                }
            }
        }
    }
    o->separated = 1; o->v2 = best; o->pen = RL(0.0);
}

/* value of one pair as a signed distance (used by the finite-difference gradient of tilted pairs) */
static real pair_value(const pose_t *pw, const real *b)
{
    pair_t o;
    pair_eval(pw, b, &o);
    return o.separated ? SQRT(o.v2) : o.pen;
}

/* scene value for one pose: d = min over boxes, with the index of the deciding box */
static real pose_distance(const pose_t *p, const real *boxes, int nb, int *arg_box)
{
    real sep2 = INF_R, pen = INF_R;
    int isep = -1, ipen = -1;
    for (int b = 0; b < nb; ++b) {
        pair_t o;
        pair_eval(p, boxes + BOX_STRIDE * (size_t)b, &o);
        if (o.separated) {
            if (o.v2 < sep2) { sep2 = o.v2; isep = b; }
        } else {
            if (o.pen < pen) { pen = o.pen; ipen = b; }
        }
    }
    if (ipen >= 0) { if (arg_box) *arg_box = ipen; return pen; }
    if (arg_box) *arg_box = isep;
    return SQRT(sep2); /* +inf for an empty scene */
}

/* partial derivatives of the pair distance w.r.t. (px,py,pz, theta[rad]) */
static void pair_grad_local(const pair_t *po, const real *b, real g[4]);

/* Tilted box: central differences of the exact pair value in (px, py, pz, theta) — the closed-form case
 * analysis of the 160 features is not worth its size for boxes no shipped scenario contains. */
#ifdef ORC_F32
#define TILT_FD_H RL(2e-3)
#define TILT_FD_COS RL(0.999998000000666666578)
#define TILT_FD_SIN RL(0.00199999866666693333)
#else
#define TILT_FD_H RL(1e-6)
#define TILT_FD_COS RL(0.9999999999995)
#define TILT_FD_SIN RL(9.99999999999833333e-07)
#endif
static void tilt_grad(const pose_t *pw, const real *b, real g[4])
{
    const real a3[3] = {pw->ax, pw->ay, pw->az};
    const real hh = TILT_FD_H, inv2h = RL(1.0) / (RL(2.0) * TILT_FD_H);
    pose_t q;
    real vp, vm;
    pose_setup_cs(&q, pw->px + hh, pw->py, pw->pz, pw->c, pw->s, a3); vp = pair_value(&q, b);
    pose_setup_cs(&q, pw->px - hh, pw->py, pw->pz, pw->c, pw->s, a3); vm = pair_value(&q, b);
    g[0] = (vp - vm) * inv2h;
    pose_setup_cs(&q, pw->px, pw->py + hh, pw->pz, pw->c, pw->s, a3); vp = pair_value(&q, b);
    pose_setup_cs(&q, pw->px, pw->py - hh, pw->pz, pw->c, pw->s, a3); vm = pair_value(&q, b);
    g[1] = (vp - vm) * inv2h;
    pose_setup_cs(&q, pw->px, pw->py, pw->pz + hh, pw->c, pw->s, a3); vp = pair_value(&q, b);
    pose_setup_cs(&q, pw->px, pw->py, pw->pz - hh, pw->c, pw->s, a3); vm = pair_value(&q, b);
    g[2] = (vp - vm) * inv2h;
    /* theta +- h: (c, s) rotated by +-h */
    real cp = FMA(-pw->s, TILT_FD_SIN, pw->c * TILT_FD_COS), sp = FMA(pw->c, TILT_FD_SIN, pw->s * TILT_FD_COS);
    real cm = FMA(pw->s, TILT_FD_SIN, pw->c * TILT_FD_COS), sm = FMA(-pw->c, TILT_FD_SIN, pw->s * TILT_FD_COS);
    pose_setup_cs(&q, pw->px, pw->py, pw->pz, cp, sp, a3); vp = pair_value(&q, b);
    pose_setup_cs(&q, pw->px, pw->py, pw->pz, cm, sm, a3); vm = pair_value(&q, b);
    g[3] = (vp - vm) * inv2h;
}

static void pair_grad(const pose_t *pw, const real *b, real g[4])
{
    if (BOX_IS_TILTED(b)) { tilt_grad(pw, b, g); return; }
    pair_t o;
    pair_eval(pw, b, &o);
    pair_grad_local(&o, b, g);
    /* the gradient came out in the box frame: rotate its xy part back to the world */
    const real cb = b[6], sb = b[7];
    real gx = cb * g[0] - sb * g[1], gy = sb * g[0] + cb * g[1];
    g[0] = gx; g[1] = gy;
}

static void pair_grad_local(const pair_t *po, const real *b, real g[4])
{
    const pair_t o = *po;
    const pose_t *p = &po->q;
    real c = p->c, s = p->s;
    real hbx = b[3], hby = b[4];
    g[0] = g[1] = g[2] = g[3] = RL(0.0);
    if (o.separated) {
        real d = SQRT(o.v2);
        if (!(d > RL(0.0))) return;
        if (o.gz > RL(0.0)) g[2] = sgn(o.dz) * o.gz / d;
        if (o.feature < 0) return;
        int f = o.feature;
        if (f < 4) {
            real rx = (f < 2) ? p->u1x : p->u2x, ry = (f < 2) ? p->u1y : p->u2y;
            if (f & 1) { rx = -rx; ry = -ry; }
            real X = o.dx + rx, Y = o.dy + ry;
            real qx = FABS(X) - hbx, qy = FABS(Y) - hby;
            real mx = qx > RL(0.0) ? sgn(X) * qx : RL(0.0);
            real my = qy > RL(0.0) ? sgn(Y) * qy : RL(0.0);
            g[0] = mx / d;
            g[1] = my / d;
            g[3] = (my * rx - mx * ry) / d;
        } else {
            real wx = (f < 6) ? o.w1x : o.w2x, wy = (f < 6) ? o.w1y : o.w2y;
            if (!(f & 1)) { wx = -wx; wy = -wy; }
            real tx = o.lx + wx, ty = o.ly + wy;
            real qx = FABS(tx) - p->ax, qy = FABS(ty) - p->ay;
            real mx = qx > RL(0.0) ? sgn(tx) * qx : RL(0.0);
            real my = qy > RL(0.0) ? sgn(ty) * qy : RL(0.0);
            g[0] = (c * mx - s * my) / d;
            g[1] = (s * mx + c * my) / d;
            g[3] = (mx * ty - my * tx) / d;
        }
    } else {
        int a = 0;
        real best = o.gx;
        if (o.gy > best) { best = o.gy; a = 1; }
        if (o.gax > best) { best = o.gax; a = 2; }
        if (o.gay > best) { best = o.gay; a = 3; }
        if (o.gz > best) { best = o.gz; a = 4; }
        real sc = sgn(c), ss = sgn(s);
        switch (a) {
        case 0: g[0] = sgn(o.dx); g[3] = -(-sc * s * p->ax + ss * c * p->ay); break;
        case 1: g[1] = sgn(o.dy); g[3] = -(ss * c * p->ax - sc * s * p->ay); break;
        case 2: { real sl = sgn(o.lx); g[0] = sl * c; g[1] = sl * s;
                  g[3] = sl * o.ly - (-sc * s * hbx + ss * c * hby); } break;
        case 3: { real sl = sgn(o.ly); g[0] = -sl * s; g[1] = sl * c;
                  g[3] = -sl * o.lx - (ss * c * hbx - sc * s * hby); } break;
        default: g[2] = sgn(o.dz); break;
        }
    }
}

/* ------------------------------------------------------------------------------
 * Point model.
 * ---------------------------------------------------------------------------- */
static real point_distance(real px, real py, real pz, const real *boxes, int nb, real radius,
                           int *arg_box)
{
    real out2 = INF_R, in = INF_R;
    int iout = -1, iin = -1;
    for (int b = 0; b < nb; ++b) {
        const real *bx = boxes + BOX_STRIDE * (size_t)b;
        real exw = px - bx[0], eyw = py - bx[1], ezw = pz - bx[2];
        real ex, ey, ez;
        if (BOX_IS_TILTED(bx)) { /* e = R^T (p - c) */
            const real *R = bx + 9;
            ex = FMA(R[6], ezw, FMA(R[3], eyw, R[0] * exw));
            ey = FMA(R[7], ezw, FMA(R[4], eyw, R[1] * exw));
            ez = FMA(R[8], ezw, FMA(R[5], eyw, R[2] * exw));
        } else {
            ex = FMA(bx[7], eyw, bx[6] * exw); ey = FMA(bx[6], eyw, -(bx[7] * exw)); ez = ezw;
        }
        real qx = FABS(ex) - bx[3], qy = FABS(ey) - bx[4], qz = FABS(ez) - bx[5];
        if (qx > RL(0.0) || qy > RL(0.0) || qz > RL(0.0)) {
            real mx = qx > RL(0.0) ? qx : RL(0.0), my = qy > RL(0.0) ? qy : RL(0.0),
                 mz = qz > RL(0.0) ? qz : RL(0.0);
            real v = FMA(mz, mz, FMA(my, my, mx * mx));
            if (v < out2) { out2 = v; iout = b; }
        } else {
            real m = maxr(maxr(qx, qy), qz);
            if (m < in) { in = m; iin = b; }
        }
    }
    if (iin >= 0) { if (arg_box) *arg_box = iin; return in; }
    if (arg_box) *arg_box = iout;
    return SQRT(out2) - radius;
}

static void point_grad(real px, real py, real pz, const real *bx, real g[3])
{
    real exw = px - bx[0], eyw = py - bx[1], ezw = pz - bx[2];
    real ex, ey, ez;
    const int tilted = BOX_IS_TILTED(bx);
    if (tilted) {
        const real *R = bx + 9;
        ex = FMA(R[6], ezw, FMA(R[3], eyw, R[0] * exw));
        ey = FMA(R[7], ezw, FMA(R[4], eyw, R[1] * exw));
        ez = FMA(R[8], ezw, FMA(R[5], eyw, R[2] * exw));
    } else {
        ex = FMA(bx[7], eyw, bx[6] * exw); ey = FMA(bx[6], eyw, -(bx[7] * exw)); ez = ezw;
    }
    real qx = FABS(ex) - bx[3], qy = FABS(ey) - bx[4], qz = FABS(ez) - bx[5];
    g[0] = g[1] = g[2] = RL(0.0);
    if (qx > RL(0.0) || qy > RL(0.0) || qz > RL(0.0)) {
        real mx = qx > RL(0.0) ? qx : RL(0.0), my = qy > RL(0.0) ? qy : RL(0.0),
             mz = qz > RL(0.0) ? qz : RL(0.0);
        real d = SQRT(FMA(mz, mz, FMA(my, my, mx * mx)));
        if (!(d > RL(0.0))) return;
        g[0] = sgn(ex) * mx / d; g[1] = sgn(ey) * my / d; g[2] = sgn(ez) * mz / d;
    } else {
        int a = 0; real best = qx;
        if (qy > best) { best = qy; a = 1; }
        if (qz > best) { best = qz; a = 2; }
        g[a] = sgn(a == 0 ? ex : a == 1 ? ey : ez);
    }
    if (tilted) { /* box frame -> world: g_w = R g_l */
        const real *R = bx + 9;
        real l0 = g[0], l1 = g[1], l2 = g[2];
        g[0] = FMA(R[2], l2, FMA(R[1], l1, R[0] * l0));
        g[1] = FMA(R[5], l2, FMA(R[4], l1, R[3] * l0));
        g[2] = FMA(R[8], l2, FMA(R[7], l1, R[6] * l0));
    } else {   /* box frame -> world */
        real gx = bx[6] * g[0] - bx[7] * g[1], gy = bx[7] * g[0] + bx[6] * g[1];
        g[0] = gx; g[1] = gy;
    }
}

/* ------------------------------------------------------------------------------
 * np.gradient(f, du) along one column and its transpose (for the reverse pass).
 * ---------------------------------------------------------------------------- */
static void np_gradient(const real *f, int n, real du, real *out)
{
    out[0] = (f[1] - f[0]) / du;
    out[n - 1] = (f[n - 1] - f[n - 2]) / du;
    for (int i = 1; i < n - 1; ++i) out[i] = (f[i + 1] - f[i - 1]) / (RL(2.0) * du);
}
static void np_gradient_T(const real *v, int n, real du, real *out)
{
    for (int i = 0; i < n; ++i) out[i] = RL(0.0);
    out[0] -= v[0] / du; out[1] += v[0] / du;
    out[n - 2] -= v[n - 1] / du; out[n - 1] += v[n - 1] / du;
    for (int k = 1; k < n - 1; ++k) {
        real t = v[k] / (RL(2.0) * du);
        out[k + 1] += t; out[k - 1] -= t;
    }
}

/* ------------------------------------------------------------------------------
 * One candidate.
 * ---------------------------------------------------------------------------- */
typedef struct {
    real P[MP_MAX_SAMPLES][3], yaw[MP_MAX_SAMPLES];
    real gP[MP_MAX_SAMPLES][3], gyaw[MP_MAX_SAMPLES];
    real d[MP_MAX_SAMPLES], gd[MP_MAX_SAMPLES];
    real d1[3][MP_MAX_SAMPLES], d2[3][MP_MAX_SAMPLES];
    real a1[3][MP_MAX_SAMPLES], a2[3][MP_MAX_SAMPLES], tmp[MP_MAX_SAMPLES];
    int arg[MP_MAX_SAMPLES];
} work_t;

static void eval_one(const mp_config *cf, const mp_task *tk, const real *boxes, const real *M,
                     const real *My, const real *tsched, const real *x, work_t *w, real *cost,
                     real *terms_out, real *grad, real *min_clear, uint8_t *flags, real *dist)
{
    const int S = cf->n_samples, C = cf->n_ctrl, Cy = cf->n_yaw_ctrl;
    const int nf = cf->n_free, ff = cf->first_free, nyf = cf->n_yaw_free;
    const int D = 3 * nf + nyf;
    const uint32_t mask = cf->term_mask;
    real W[MP_MAX_CTRL][3], Y[MP_MAX_CTRL], term[MP_NTERMS];
    for (int j = 0; j < C; ++j)
        for (int a = 0; a < 3; ++a) W[j][a] = (real)tk->ctrl[j][a];
    for (int j = 0; j < nf; ++j)
        for (int a = 0; a < 3; ++a) W[ff + j][a] = x[3 * j + a];
    for (int j = 0; j < Cy; ++j) Y[j] = (real)tk->yaw_ctrl[j];
    for (int j = 0; j < nyf; ++j) Y[1 + j] = x[3 * nf + j];
    for (int t = 0; t < MP_NTERMS; ++t) term[t] = RL(0.0);

    /* spline samples: Q = M W, fixed fma chain over j (spline_opt.py:511-514) */
    for (int s = 0; s < S; ++s) {
        for (int a = 0; a < 3; ++a) {
            real acc = M[s * C] * W[0][a];
            for (int j = 1; j < C; ++j) acc = FMA(M[s * C + j], W[j][a], acc);
            w->P[s][a] = acc;
            w->gP[s][a] = RL(0.0);
        }
        real acc = My[s * Cy] * Y[0];
        for (int j = 1; j < Cy; ++j) acc = FMA(My[s * Cy + j], Y[j], acc);
        w->yaw[s] = acc;
        w->gyaw[s] = RL(0.0);
    }
    const real *wt = NULL;
    real wts[MP_NTERMS];
    for (int t = 0; t < MP_NTERMS; ++t) wts[t] = (real)cf->weights[t];
    wt = wts;

    /* --- path_length (:54-60) --- */
    if (mask & (1u << MP_T_LEN)) {
        real acc = RL(0.0);
        for (int i = 0; i + 1 < S; ++i) {
            real ex = w->P[i + 1][0] - w->P[i][0], ey = w->P[i + 1][1] - w->P[i][1],
                 ez = w->P[i + 1][2] - w->P[i][2];
            real n = SQRT(ex * ex + ey * ey + ez * ez);
            acc += n;
            if (grad && n > RL(0.0)) {
                real k = wt[MP_T_LEN] / n;
                w->gP[i + 1][0] += k * ex; w->gP[i + 1][1] += k * ey; w->gP[i + 1][2] += k * ez;
                w->gP[i][0] -= k * ex; w->gP[i][1] -= k * ey; w->gP[i][2] -= k * ez;
            }
        }
        term[MP_T_LEN] = acc;
    }

    /* --- curvature_cost (:63-79) --- */
    if ((mask & (1u << MP_T_CURV)) && S >= 3) {
        const real du = RL(1.0) / (real)(S - 1);
        const real eps = RL(1e-9);
        for (int a = 0; a < 3; ++a) {
            for (int i = 0; i < S; ++i) w->tmp[i] = w->P[i][a];
            np_gradient(w->tmp, S, du, w->d1[a]);
            np_gradient(w->d1[a], S, du, w->d2[a]);
        }
        real acc = RL(0.0);
        for (int i = 0; i < S; ++i) {
            real v1x = w->d1[0][i], v1y = w->d1[1][i], v1z = w->d1[2][i];
            real v2x = w->d2[0][i], v2y = w->d2[1][i], v2z = w->d2[2][i];
            real speed = SQRT(v1x * v1x + v1y * v1y + v1z * v1z);
            real cx = v1y * v2z - v1z * v2y, cy = v1z * v2x - v1x * v2z, cz = v1x * v2y - v1y * v2x;
            real cn = SQRT(cx * cx + cy * cy + cz * cz);
            real sp = speed > eps ? speed : eps;
            real sp3 = sp * sp * sp;
            real kappa = cn / sp3;
            acc += (kappa * kappa) * speed;
            if (grad) {
                /* f = cn^2 * speed / sp^6 */
                real sp6 = sp3 * sp3;
                real k = wt[MP_T_CURV] * du;
                real gcx = k * RL(2.0) * cx * speed / sp6, gcy = k * RL(2.0) * cy * speed / sp6,
                     gcz = k * RL(2.0) * cz * speed / sp6;
                real gs = cn * cn / sp6; /* d f / d speed (explicit) */
                if (speed > eps) gs += RL(-6.0) * cn * cn * speed / (sp6 * sp);
                gs *= k;
                real g1x = v2y * gcz - v2z * gcy, g1y = v2z * gcx - v2x * gcz,
                     g1z = v2x * gcy - v2y * gcx;
                real g2x = gcy * v1z - gcz * v1y, g2y = gcz * v1x - gcx * v1z,
                     g2z = gcx * v1y - gcy * v1x;
                if (speed > RL(0.0)) {
                    g1x += gs * v1x / speed; g1y += gs * v1y / speed; g1z += gs * v1z / speed;
                }
                w->a1[0][i] = g1x; w->a1[1][i] = g1y; w->a1[2][i] = g1z;
                w->a2[0][i] = g2x; w->a2[1][i] = g2y; w->a2[2][i] = g2z;
            }
        }
        term[MP_T_CURV] = acc * du;
        if (grad) {
            for (int a = 0; a < 3; ++a) {
                np_gradient_T(w->a2[a], S, du, w->tmp);
                for (int i = 0; i < S; ++i) w->a1[a][i] += w->tmp[i];
                np_gradient_T(w->a1[a], S, du, w->tmp);
                for (int i = 0; i < S; ++i) w->gP[i][a] += w->tmp[i];
            }
        }
    }

    /* --- yaw_smoothness_cost (:100-106) --- */
    if ((mask & (1u << MP_T_YAW_SMOOTH)) && S >= 3) {
        real acc = RL(0.0);
        for (int i = 0; i + 2 < S; ++i) {
            real e = w->yaw[i] - RL(2.0) * w->yaw[i + 1] + w->yaw[i + 2];
            acc += e * e;
            if (grad) {
                real k = wt[MP_T_YAW_SMOOTH] * RL(2.0) * e;
                w->gyaw[i] += k; w->gyaw[i + 1] -= RL(2.0) * k; w->gyaw[i + 2] += k;
            }
        }
        term[MP_T_YAW_SMOOTH] = acc;
    }

    /* --- clearance profile d_safe (:538-545, _path_distances :148-186) --- */
    int Sp = cf->n_safe;
    const int nb = tk->box_count;
    const real *bx = boxes + BOX_STRIDE * (size_t)tk->box_begin;
    real a3[3] = {(real)cf->half_extents[0], (real)cf->half_extents[1], (real)cf->half_extents[2]};
    for (int k = 0; k < Sp; ++k) {
        int s = cf->idx_safe[k];
        if (cf->model == MP_MODEL_BOX) {
            pose_t ps;
            pose_setup(&ps, w->P[s][0], w->P[s][1], w->P[s][2], w->yaw[s], a3);
            w->d[k] = pose_distance(&ps, bx, nb, &w->arg[k]);
        } else {
            w->d[k] = point_distance(w->P[s][0], w->P[s][1], w->P[s][2], bx, nb,
                                     (real)cf->point_radius, &w->arg[k]);
        }
        w->gd[k] = RL(0.0);
        if (dist) dist[k] = w->d[k];
    }
    if (cf->last_if_not_close && Sp >= 2) {
        /* P[::k] then append P[-1] unless np.allclose(P_safe[-1], P[-1]) (:802-805) */
        int sa = cf->idx_safe[Sp - 2], sb = S - 1, close = 1;
        for (int a = 0; a < 3; ++a)
            if (!(FABS(w->P[sa][a] - w->P[sb][a]) <= RL(1e-8) + RL(1e-5) * FABS(w->P[sb][a]))) close = 0;
        if (close) Sp -= 1;
    }
    const real req = (real)cf->required_clearance, pref = (real)cf->preferred_clearance;
    const real *d = w->d;
    real *gd = w->gd;

    if (mask & (1u << MP_T_SAFE)) { /* :546-547 */
        real acc = RL(0.0);
        for (int k = 0; k < Sp; ++k) {
            real e = npmax0(req - d[k]);
            acc += e * e;
            gd[k] -= wt[MP_T_SAFE] * RL(2.0) * e;
        }
        term[MP_T_SAFE] = acc;
    }
    if ((mask & (1u << MP_T_SAFE_PREF)) && cf->use_pref) { /* :548-556 */
        real acc = RL(0.0);
        for (int k = 0; k < cf->keep_n && k < Sp; ++k) {
            real e = npmax0(pref - d[k]);
            acc += e * e;
            gd[k] -= wt[MP_T_SAFE_PREF] * RL(2.0) * e;
        }
        term[MP_T_SAFE_PREF] = acc;
    }
    if (mask & (1u << MP_T_REBOUND)) { /* :559-562 */
        real acc = RL(0.0);
        for (int k = cf->tail_start; k + 1 < Sp; ++k) {
            real e = npmax0(d[k + 1] - d[k]);
            acc += e * e;
            gd[k + 1] += wt[MP_T_REBOUND] * RL(2.0) * e;
            gd[k] -= wt[MP_T_REBOUND] * RL(2.0) * e;
        }
        term[MP_T_REBOUND] = acc;
    }
    if (Sp >= 1) {
        real endc = d[Sp - 1];
        if (mask & (1u << MP_T_GOAL_CLEAR)) { /* :565-566 */
            real e = endc - pref;
            e = e > RL(0.0) ? e : RL(0.0); /* python max(0.0, e): NaN -> 0.0 */
            term[MP_T_GOAL_CLEAR] = e * e;
            gd[Sp - 1] += wt[MP_T_GOAL_CLEAR] * RL(2.0) * e;
        }
        if ((mask & (1u << MP_T_GOAL_TARGET)) && cf->use_goal_target) { /* :567-569 */
            real e = endc - (real)cf->goal_clearance_target;
            term[MP_T_GOAL_TARGET] = e * e;
            gd[Sp - 1] += wt[MP_T_GOAL_TARGET] * RL(2.0) * e;
        }
    }
    if (mask & ((1u << MP_T_APPROACH_CLEAR) | (1u << MP_T_APPROACH_COL))) { /* :572-581 */
        real acc1 = RL(0.0), acc2 = RL(0.0);
        real tgt = (real)cf->approach_target;
        for (int k = 0; k < Sp; ++k) {
            if (!cf->approach_mask[k]) continue;
            real e1 = npmax0(tgt - d[k]);
            real e2 = npmax0(-d[k]);
            if (mask & (1u << MP_T_APPROACH_CLEAR)) { acc1 += e1 * e1; gd[k] -= wt[MP_T_APPROACH_CLEAR] * RL(2.0) * e1; }
            if (mask & (1u << MP_T_APPROACH_COL)) { acc2 += e2 * e2; gd[k] -= wt[MP_T_APPROACH_COL] * RL(2.0) * e2; }
        }
        term[MP_T_APPROACH_CLEAR] = acc1;
        term[MP_T_APPROACH_COL] = acc2;
    }

    /* --- control-point deviations (:584-585) --- */
    if (mask & (1u << MP_T_VIA_DEV)) {
        real acc = RL(0.0);
        for (int j = 0; j < nf; ++j)
            for (int a = 0; a < 3; ++a) {
                real e = W[ff + j][a] - (real)tk->ctrl[ff + j][a];
                acc += e * e;
            }
        term[MP_T_VIA_DEV] = acc;
    }
    if (mask & (1u << MP_T_YAW_DEV)) {
        real acc = RL(0.0);
        for (int j = 0; j < Cy; ++j) {
            real e = Y[j] - (real)tk->yaw_ctrl[j];
            acc += e * e;
        }
        term[MP_T_YAW_DEV] = acc;
    }
    /* --- yaw monotonicity (:588-593) --- */
    if (mask & (1u << MP_T_YAW_MONO)) {
        real acc = RL(0.0);
        for (int i = 0; i + 1 < S; ++i) {
            real dy = w->yaw[i + 1] - w->yaw[i];
            real e = npmax0(tk->yaw_increasing ? -dy : dy);
            acc += e * e;
            if (grad) {
                real k = wt[MP_T_YAW_MONO] * RL(2.0) * e * (tk->yaw_increasing ? RL(-1.0) : RL(1.0));
                w->gyaw[i + 1] += k; w->gyaw[i] -= k;
            }
        }
        term[MP_T_YAW_MONO] = acc;
    }
    /* --- yaw schedule (:596-598) --- */
    if (mask & (1u << MP_T_YAW_SCHED)) {
        real y0 = (real)tk->yaw_ctrl[0], y1 = (real)tk->yaw_ctrl[Cy - 1];
        real acc = RL(0.0);
        for (int i = 0; i < S; ++i) {
            real target = y0 + (y1 - y0) * tsched[i];
            real e = w->yaw[i] - target;
            acc += e * e;
            if (grad) w->gyaw[i] += wt[MP_T_YAW_SCHED] * RL(2.0) * e;
        }
        term[MP_T_YAW_SCHED] = acc;
    }
    /* --- goal approach alignment (:117-145) --- */
    if ((mask & (1u << MP_T_GOAL_NORMAL)) && tk->has_goal_normal && S >= 3) {
        int first = S - cf->goal_tail_n;
        if (first < 0) first = 0;
        real vx = RL(0.0), vy = RL(0.0), vz = RL(0.0);
        for (int i = first; i + 1 < S; ++i) { /* np.sum(np.diff(P_tail)) */
            vx += w->P[i + 1][0] - w->P[i][0];
            vy += w->P[i + 1][1] - w->P[i][1];
            vz += w->P[i + 1][2] - w->P[i][2];
        }
        real n = SQRT(vx * vx + vy * vy + vz * vz);
        if (first < S - 1 && !(n < RL(1e-12))) {
            real ux = vx / n, uy = vy / n, uz = vz / n;
            if (ux != RL(0.0) || uy != RL(0.0) || uz != RL(0.0)) {
                real sx = (real)tk->goal_normal[0], sy = (real)tk->goal_normal[1],
                     sz = (real)tk->goal_normal[2];
                real c = ux * -sx + uy * -sy + uz * -sz;
                real cc = c < RL(-1.0) ? RL(-1.0) : (c > RL(1.0) ? RL(1.0) : c);
                real e = RL(1.0) - cc;
                term[MP_T_GOAL_NORMAL] = e * e;
                if (grad && c > RL(-1.0) && c < RL(1.0)) {
                    /* dc/dv = -(s - u (u.s)) / n */
                    real us = ux * sx + uy * sy + uz * sz;
                    real k = wt[MP_T_GOAL_NORMAL] * RL(-2.0) * e;
                    real gx = k * -(sx - ux * us) / n, gy = k * -(sy - uy * us) / n,
                         gz = k * -(sz - uz * us) / n;
                    w->gP[S - 1][0] += gx; w->gP[S - 1][1] += gy; w->gP[S - 1][2] += gz;
                    w->gP[first][0] -= gx; w->gP[first][1] -= gy; w->gP[first][2] -= gz;
                }
            }
        }
    }

    /* --- weighted sum in the reference's order (:605-621) --- */
    real j = RL(0.0);
    for (int t = 0; t < MP_NTERMS; ++t)
        if (mask & (1u << t)) j = (t == 0) ? wt[t] * term[t] : j + wt[t] * term[t];
    *cost = j;
    if (terms_out)
        for (int t = 0; t < MP_NTERMS; ++t) terms_out[t] = term[t];

    /* --- min clearance and flags --- */
    real mc = INF_R;
    for (int k = 0; k < Sp; ++k)
        if (d[k] < mc) mc = d[k];
    if (min_clear) *min_clear = mc;
    if (flags) *flags = (uint8_t)((mc < RL(0.0) ? MP_FLAG_COLLISION : 0u) | (mc >= req ? MP_FLAG_FEASIBLE : 0u));

    /* --- gradient: clearance chain, then x = M^T g --- */
    if (grad) {
        const real DEG2RAD = RL(0.017453292519943295);
        for (int k = 0; k < Sp; ++k) {
            if (gd[k] == RL(0.0) || w->arg[k] < 0) continue;
            int s = cf->idx_safe[k];
            const real *bb = bx + BOX_STRIDE * (size_t)w->arg[k];
            if (cf->model == MP_MODEL_BOX) {
                pose_t ps;
                real g[4];
                pose_setup(&ps, w->P[s][0], w->P[s][1], w->P[s][2], w->yaw[s], a3);
                pair_grad(&ps, bb, g);
                w->gP[s][0] += gd[k] * g[0]; w->gP[s][1] += gd[k] * g[1]; w->gP[s][2] += gd[k] * g[2];
                w->gyaw[s] += gd[k] * g[3] * DEG2RAD;
            } else {
                real g[3];
                point_grad(w->P[s][0], w->P[s][1], w->P[s][2], bb, g);
                w->gP[s][0] += gd[k] * g[0]; w->gP[s][1] += gd[k] * g[1]; w->gP[s][2] += gd[k] * g[2];
            }
        }
        for (int i = 0; i < D; ++i) grad[i] = RL(0.0);
        for (int jv = 0; jv < nf; ++jv)
            for (int a = 0; a < 3; ++a) {
                real acc = RL(0.0);
                for (int s = 0; s < S; ++s) acc += M[s * C + ff + jv] * w->gP[s][a];
                if (mask & (1u << MP_T_VIA_DEV))
                    acc += wt[MP_T_VIA_DEV] * RL(2.0) * (W[ff + jv][a] - (real)tk->ctrl[ff + jv][a]);
                grad[3 * jv + a] = acc;
            }
        for (int jv = 0; jv < nyf; ++jv) {
            real acc = RL(0.0);
            for (int s = 0; s < S; ++s) acc += My[s * Cy + 1 + jv] * w->gyaw[s];
            if (mask & (1u << MP_T_YAW_DEV))
                acc += wt[MP_T_YAW_DEV] * RL(2.0) * (Y[1 + jv] - (real)tk->yaw_ctrl[1 + jv]);
            grad[3 * nf + jv] = acc;
        }
    }
}

/* ------------------------------------------------------------------------------
 * Exported entry points (ctypes).  boxes8: [n_boxes x 18] doubles = cx,cy,cz, hx,hy,hz, cos,sin (sin = 2: tilted), spare, R[9] (row major, tilted boxes only).
 * cfg->basis / basis_yaw / t_sched must be non-NULL.
 * ---------------------------------------------------------------------------- */
static real *to_real(const double *src, size_t n)
{
    real *out = (real *)malloc((n ? n : 1) * sizeof(real));
    for (size_t i = 0; i < n; ++i) out[i] = (real)src[i];
    return out;
}

int ORC_NAME(orc_eval)(const mp_config *cf, const mp_task *tasks, int32_t n_tasks,
                       int64_t cands_per_task, const int32_t *task_of, int32_t n_boxes,
                       const double *boxes8, int64_t n, const real *X, real *cost, real *terms,
                       real *grad, real *min_clear, uint8_t *flags, real *dist)
{
    if (!cf || !tasks || !cf->basis || !cf->basis_yaw || !cf->t_sched) return MP_E_ARG;
    if (cf->n_samples > MP_MAX_SAMPLES || cf->n_ctrl > MP_MAX_CTRL) return MP_E_ARG;
    const int S = cf->n_samples, C = cf->n_ctrl, Cy = cf->n_yaw_ctrl;
    const int D = 3 * cf->n_free + cf->n_yaw_free;
    real *bx = to_real(boxes8, BOX_STRIDE * (size_t)n_boxes);
    real *M = to_real(cf->basis, (size_t)S * C);
    real *My = to_real(cf->basis_yaw, (size_t)S * Cy);
    real *ts = to_real(cf->t_sched, (size_t)S);
    work_t *w = (work_t *)malloc(sizeof(work_t));
    int rc = MP_OK;
    for (int64_t i = 0; i < n; ++i) {
        int64_t t = task_of ? task_of[i] : (cands_per_task > 0 ? i / cands_per_task : 0);
        if (t < 0 || t >= n_tasks) { rc = MP_E_ARG; break; }
        eval_one(cf, tasks + t, bx, M, My, ts, X + i * D, w, cost + i,
                 terms ? terms + i * MP_NTERMS : NULL, grad ? grad + i * D : NULL,
                 min_clear ? min_clear + i : NULL, flags ? flags + i : NULL,
                 dist ? dist + i * cf->n_safe : NULL);
    }
    free(w); free(ts); free(My); free(M); free(bx);
    return rc;
}

int ORC_NAME(orc_point_sdf)(int32_t n_boxes, const double *boxes8, double radius, int64_t n,
                            const real *pts, real *out)
{
    real *bx = to_real(boxes8, BOX_STRIDE * (size_t)n_boxes);
    for (int64_t i = 0; i < n; ++i)
        out[i] = point_distance(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], bx, n_boxes, (real)radius, NULL);
    free(bx);
    return MP_OK;
}

int ORC_NAME(orc_box_sdf)(int32_t n_boxes, const double *boxes8, const double size[3], int64_t n,
                          const real *pos, const real *yaw_deg, real *out)
{
    real *bx = to_real(boxes8, BOX_STRIDE * (size_t)n_boxes);
    real a3[3] = {(real)(0.5 * size[0]), (real)(0.5 * size[1]), (real)(0.5 * size[2])};
    for (int64_t i = 0; i < n; ++i) {
        pose_t ps;
        pose_setup(&ps, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], yaw_deg[i], a3);
        out[i] = pose_distance(&ps, bx, n_boxes, NULL);
    }
    free(bx);
    return MP_OK;
}

/* same query with the rotation given as (cos, sin): used by the golden generator's stub
 * scene, which receives the reference's own quaternion (scene.py:70-76). */
int ORC_NAME(orc_box_sdf_cs)(int32_t n_boxes, const double *boxes8, const double size[3], int64_t n,
                             const real *pos, const real *c, const real *s, real *out)
{
    real *bx = to_real(boxes8, BOX_STRIDE * (size_t)n_boxes);
    real a3[3] = {(real)(0.5 * size[0]), (real)(0.5 * size[1]), (real)(0.5 * size[2])};
    for (int64_t i = 0; i < n; ++i) {
        pose_t ps;
        pose_setup_cs(&ps, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], c[i], s[i], a3);
        out[i] = pose_distance(&ps, bx, n_boxes, NULL);
    }
    free(bx);
    return MP_OK;
}

/* (cos yaw, sin yaw) as composed above — lets the tests pin the trig against the
 * reference's yaw_deg_to_quat + quat_to_rot. */
int ORC_NAME(orc_yaw_rot)(int64_t n, const real *yaw_deg, real *c, real *s)
{
    for (int64_t i = 0; i < n; ++i) orc_yaw_rot(yaw_deg[i], c + i, s + i);
    return MP_OK;
}
