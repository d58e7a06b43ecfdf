"""Planner parameter sets the reference ships with its scripts (values only).

BASE_CONFIG: example/benchmark_planners.py:30-61.  DEMO_CONFIG adds the demo's solver
choice (example/demo.py:14-47: same weights, method Powell, maxiter 160)."""

BASE_CONFIG = dict(
    n_vias=2, n_yaw_vias=2, combined_4d=True,
    n_samples_curve=101, collision_check_subsample=1,
    safety_margin=0.0, preferred_safety_margin=0.02, relax_preferred_final_fraction=0.25,
    approach_only_clearance=0.015, contact_window_fraction=0.08, approach_fraction=0.25,
    goal_clearance_target=0.0, yaw_goal_reach_u=0.5, goal_approach_window_fraction=0.12,
    init_offset_scale=0.7,
    w_len=5.0, w_curv=0.12, w_yaw_smooth=0.008, w_safe=380.0, w_safe_preferred=24.0,
    w_approach_rebound=280.0, w_goal_clearance=35.0, w_goal_clearance_target=260.0,
    w_approach_clearance=420.0, w_approach_collision=1400.0, w_via_dev=0.06, w_yaw_dev=0.05,
    w_yaw_monotonic=80.0, w_yaw_schedule=55.0, w_goal_approach_normal=80.0,
)

DEMO_CONFIG = dict(BASE_CONFIG, method="Powell", options={"maxiter": 160, "xtol": 1e-3, "ftol": 1e-3})
