/*
 * Plain-C caller of libmp_hip.so: shows that the boundary of include/mp_api.h needs nothing but C
 * (no Python, no torch).  Builds a 4-waypoint, 16-sample problem against two boxes, evaluates three
 * candidates with host buffers and prints cost / min clearance / flags / gradient, one line each;
 * tests/test_gpu_parity.py::test_plain_c_caller compares the lines with the Python host + oracle.
 *
 *   gcc -I include tests/cabi/cabi_demo.c -o tests/cabi/cabi_demo -L motion_planning_amd -lmp_hip -Wl,-rpath,...
 */
#include <stdio.h>
#include <string.h>

#include "mp_api.h"

#define S 16
#define C4 4

int main(void)
{
    mp_ctx *ctx = NULL;
    int rc = mp_create(0, MP_F64, &ctx);
    if (rc) { fprintf(stderr, "mp_create: %s\n", mp_last_error(NULL)); return 2; }

    const double cx[2] = {0.0, 0.6}, cy[2] = {0.0, 0.1}, cz[2] = {0.05, 0.5};
    const double hx[2] = {2.0, 0.2}, hy[2] = {2.0, 0.3}, hz[2] = {0.05, 0.4};
    if ((rc = mp_set_scene(ctx, 2, cx, cy, cz, hx, hy, hz, NULL))) { fprintf(stderr, "%s\n", mp_last_error(ctx)); return 2; }

    static int32_t idx[S];
    static uint8_t mask[S];
    static double tsched[S];
    for (int i = 0; i < S; ++i) {
        idx[i] = i;
        mask[i] = (uint8_t)(i < S - 2);
        tsched[i] = (double)i / (S - 1);
    }
    mp_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_samples = S; cfg.n_ctrl = C4; cfg.first_free = 1; cfg.n_free = 2; cfg.n_yaw_ctrl = C4; cfg.n_yaw_free = 2;
    cfg.model = MP_MODEL_BOX; cfg.n_safe = S; cfg.keep_n = S; cfg.use_pref = 1; cfg.tail_start = S - 4;
    cfg.use_goal_target = 1; cfg.goal_tail_n = 3; cfg.term_mask = 0x7fff;
    for (int t = 0; t < MP_NTERMS; ++t) cfg.weights[t] = 1.0 + 0.25 * t;
    cfg.required_clearance = 0.0; cfg.preferred_clearance = 0.05; cfg.approach_target = 0.03;
    cfg.goal_clearance_target = 0.0; cfg.point_radius = 1e-6;
    cfg.half_extents[0] = 0.15; cfg.half_extents[1] = 0.1; cfg.half_extents[2] = 0.1;
    cfg.idx_safe = idx; cfg.approach_mask = mask; cfg.t_sched = tsched;
    cfg.basis = NULL; cfg.basis_yaw = NULL; /* let the library build the not-a-knot sampling matrices */

    mp_task task;
    memset(&task, 0, sizeof task);
    const double wp[C4][3] = {{-1.0, -0.5, 1.0}, {-0.4, -0.2, 0.9}, {0.2, 0.1, 0.8}, {1.0, 0.4, 0.3}};
    const double yaw[C4] = {0.0, 20.0, 40.0, 60.0};
    for (int j = 0; j < C4; ++j) {
        for (int a = 0; a < 3; ++a) task.ctrl[j][a] = wp[j][a];
        task.yaw_ctrl[j] = yaw[j];
    }
    task.goal_normal[2] = 1.0; task.has_goal_normal = 1; task.yaw_increasing = 1; task.box_begin = 0; task.box_count = 2;
    if ((rc = mp_set_problem(ctx, &cfg, 1, &task))) { fprintf(stderr, "%s\n", mp_last_error(ctx)); return 2; }

    const double X[3][8] = {
        {-0.4, -0.2, 0.9, 0.2, 0.1, 0.8, 20.0, 40.0},
        {-0.3, 0.3, 1.2, 0.5, 0.2, 0.6, 10.0, 70.0},
        {0.1, -0.1, 0.5, 0.6, 0.1, 0.55, 35.0, 30.0},
    };
    double cost[3], minc[3], grad[3][8], terms[3][MP_NTERMS];
    uint8_t flags[3];
    if ((rc = mp_eval(ctx, 3, X, NULL, cost, terms, grad, minc, flags, NULL, MP_MEM_HOST))) {
        fprintf(stderr, "%s\n", mp_last_error(ctx));
        return 2;
    }
    for (int i = 0; i < 3; ++i) {
        printf("%.17g %.17g %u", cost[i], minc[i], (unsigned)flags[i]);
        for (int d = 0; d < 8; ++d) printf(" %.17g", grad[i][d]);
        printf("\n");
    }
    /* error path: a second problem whose task points outside the scene must be refused */
    task.box_count = 3;
    rc = mp_set_problem(ctx, &cfg, 1, &task);
    printf("bad_task rc=%d\n", rc);
    mp_destroy(ctx);
    return 0;
}
