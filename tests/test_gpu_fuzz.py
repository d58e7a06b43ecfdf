"""Randomised parity: planner configurations drawn at random (sample counts that are not multiples of the wave, up to
14 vias, separate yaw spline, subsampled collision checks, every weight on or off, rotated and tilted static boxes, empty-ish and
cluttered scenes), HIP path through the C ABI against the CPU oracle.  Same bar as tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest

from motion_planning_amd import _abi as abi
from motion_planning_amd.problem import build_path_problem, build_vias_problem
from motion_planning_amd.synthetic import random_scene
from tests import helpers as H
from tests.test_gpu_parity import _compare

pytestmark = pytest.mark.gpu


def _random_path_problem(rng, basis_fn):
    start = rng.uniform([-1.8, -1.8, 0.6], [-0.8, -0.8, 1.8])
    goal = rng.uniform([0.8, 0.8, 0.35], [1.8, 1.8, 1.2])
    n_boxes = int(rng.choice([1, 2, 5, 17, 50, 140]))
    scene = random_scene(n_boxes, int(rng.integers(1 << 30)), (0.05, 0.5), start, goal)
    w = lambda p=0.7, hi=5.0: float(rng.uniform(0.0, hi)) if rng.random() < p else 0.0
    combined = bool(rng.random() < 0.6)
    n_vias = int(rng.choice([2, 2, 3, 4, 6, 9, 14]))
    kw = dict(
        n_vias=n_vias, n_samples_curve=int(rng.choice([16, 33, 64, 65, 101, 127, 128, 200])),
        collision_check_subsample=int(rng.choice([1, 1, 2, 3, 5])),
        start_yaw_deg=float(rng.uniform(-180, 180)), goal_yaw_deg=float(rng.uniform(-180, 180)),
        combined_4d=combined, n_yaw_vias=n_vias if combined else int(rng.choice([0, 1, 2, 3])),
        safety_margin=float(rng.uniform(0.0, 0.05)),
        preferred_safety_margin=float(rng.uniform(0.0, 0.15)) if rng.random() < 0.7 else None,
        relax_preferred_final_fraction=float(rng.choice([0.0, 0.25, 0.5])),
        w_len=w(0.9), w_curv=w(0.8, 0.5), w_yaw_smooth=w(0.5, 0.1), w_safe=w(0.9, 80.0), w_safe_preferred=w(0.6, 10.0),
        w_approach_rebound=w(0.5, 10.0), w_goal_clearance=w(0.5, 10.0),
        goal_clearance_target=float(rng.uniform(0.0, 0.05)) if rng.random() < 0.5 else None, w_goal_clearance_target=w(0.5, 10.0),
        approach_only_clearance=float(rng.uniform(0.0, 0.1)) if rng.random() < 0.5 else None,
        contact_window_fraction=float(rng.uniform(0.02, 0.3)), w_approach_clearance=w(0.5, 10.0), w_approach_collision=w(0.5, 50.0),
        approach_fraction=float(rng.uniform(0.05, 0.5)), w_via_dev=w(0.5, 1.0), w_yaw_dev=w(0.5, 0.01),
        w_yaw_monotonic=w(0.5, 1.0), yaw_goal_reach_u=float(rng.uniform(0.3, 1.0)), w_yaw_schedule=w(0.5, 0.01),
        goal_approach_normals=rng.normal(size=(int(rng.integers(1, 4)), 3)) if rng.random() < 0.7 else None,
        goal_approach_window_fraction=float(rng.uniform(0.05, 0.3)), w_goal_approach_normal=w(0.6, 5.0),
        init_offset_scale=float(rng.uniform(0.3, 1.2)),
    )
    if not combined and kw["n_yaw_vias"] == 0:
        kw["n_yaw_vias"] = 0
    box_model = rng.random() < 0.8
    if box_model:
        kw["moving_block_size"] = tuple(rng.uniform(0.1, 0.9, 3))
    prob = build_path_problem(scene, start, goal, basis_fn, **kw)
    u = rng.random()
    if u < 0.3:
        prob = H.with_rotated_statics(prob, seed=int(rng.integers(1 << 30)))
    elif u < 0.5:
        prob = H.with_tilted_statics(prob, seed=int(rng.integers(1 << 30)), fraction=float(rng.choice([0.1, 0.5, 1.0])))
    return prob


@pytest.mark.parametrize("seed", range(int(os.environ.get("MP_FUZZ_SEEDS", "36"))))   # MP_FUZZ_SEEDS=500: a longer one-off run
def test_random_configuration_matches_oracle(seed, oracle_lib):
    from motion_planning_amd import _native as native

    rng = np.random.default_rng(1000 + seed)
    prob = _random_path_problem(rng, native.build_basis)
    n = int(rng.choice([7, 64, 130]))
    X = prob.x0 + prob.sigma0 * rng.normal(size=(n, prob.dim)) * float(rng.choice([0.3, 1.0, 2.0]))
    dt = np.float32 if seed % 6 == 5 else np.float64
    ctx = native.Context(0, dt)
    ctx.set_problem(prob)
    Xd = np.ascontiguousarray(X, dtype=dt)
    out = ctx.eval(Xd, want_grad=True, want_dist=True)
    ref = oracle_lib.evaluate(prob, Xd, dtype=dt)
    _compare(out, ref, dt)
    out2 = ctx.eval(Xd, want_grad=False, want_terms=False)
    np.testing.assert_array_equal(out2["flags"], ref["flags"])
    np.testing.assert_array_equal(out2["min_clear"], ref["min_clear"])
    out3 = ctx.eval(Xd, want_grad=True, want_terms=False)      # the launch DeviceBatch makes: no term breakdown
    H.assert_same_gradient(out3["grad"], out["grad"], prob, dt)
    np.testing.assert_array_equal(out3["flags"], ref["flags"])
    ctx.close()


@pytest.mark.parametrize("seed", range(8))
def test_random_vias_configuration_matches_oracle(seed, oracle_lib):
    from motion_planning_amd import _native as native

    rng = np.random.default_rng(5000 + seed)
    start = rng.uniform([-1.8, -1.8, 0.6], [-0.8, -0.8, 1.8])
    goal = rng.uniform([0.8, 0.8, 0.35], [1.8, 1.8, 1.2])
    via = 0.5 * (start + goal) + rng.normal(size=3) * 0.3
    scene = random_scene(int(rng.choice([1, 4, 30])), int(rng.integers(1 << 30)), (0.05, 0.5), start, goal)
    kw = dict(n_additional_vias=int(rng.choice([1, 2, 5, 10])), n_samples_curve=int(rng.choice([31, 61, 121, 150])),
              collision_check_subsample=int(rng.choice([1, 2, 5])), safety_margin=float(rng.uniform(0.0, 0.1)),
              w_len=float(rng.uniform(0, 2)), w_curv=float(rng.uniform(0, 0.5)), w_safe=float(rng.uniform(0, 80)))
    if rng.random() < 0.5:
        kw["tool_half_extents"] = tuple(rng.uniform(0.05, 0.3, 3))   # fixed-orientation box model (:784-789)
    prob = build_vias_problem(scene, start, via, goal, native.build_basis, **kw)
    X = prob.x0 + prob.sigma0 * rng.normal(size=(50, prob.dim))
    ctx = native.Context(0)
    ctx.set_problem(prob)
    out = ctx.eval(X, want_grad=True, want_dist=True)
    ref = oracle_lib.evaluate(prob, X)
    _compare(out, ref, np.float64)
    ctx.close()


@pytest.mark.parametrize("S", [256, 400, abi.MP_MAX_SAMPLES])
def test_maximum_sizes(S, oracle_lib):
    """Largest supported trajectories: 16 waypoints, up to MP_MAX_SAMPLES samples (an LDS slice beyond 64 KB per wave in
    the fused gradient kernel), fused and split clearance."""
    import os

    from motion_planning_amd import _native as native

    rng = np.random.default_rng(S)
    start, goal = np.array([-1.5, -1.5, 1.5]), np.array([1.5, 1.5, 0.4])
    scene = random_scene(40, 11, (0.1, 0.5), start, goal)
    prob = build_path_problem(scene, start, goal, native.build_basis, n_vias=14, n_samples_curve=S, moving_block_size=(0.9, 0.6, 0.6),
                              start_yaw_deg=0.0, goal_yaw_deg=90.0, n_yaw_vias=14, w_curv=0.1, w_safe=50.0, w_safe_preferred=5.0,
                              preferred_safety_margin=0.05, w_yaw_monotonic=1.0, w_approach_rebound=1.0)
    assert prob.n_ctrl == abi.MP_MAX_CTRL
    X = prob.x0 + 0.3 * prob.sigma0 * rng.normal(size=(9, prob.dim))
    ref = oracle_lib.evaluate(prob, X)
    for force in ("0", "1"):
        os.environ["MP_SPLIT_CLEARANCE"] = force
        try:
            ctx = native.Context(0)
            ctx.set_problem(prob)
            # a gradient launch keeps 15 fields + 4 bytes per sample and wave in LDS: MP_MAX_SAMPLES = 1024 fp64 samples are a
            # 125 KB slice, one wave per CU (round 1: 22 arrays, gradient launches beyond ~860 samples were refused)
            out = ctx.eval(X, want_grad=True, want_dist=True)
            _compare(out, ref, np.float64)
            out = ctx.eval(X, want_grad=False, want_terms=True, want_dist=True)
            np.testing.assert_array_equal(out["dist"], ref["dist"])
            np.testing.assert_array_equal(out["flags"], ref["flags"])
            assert H.rel_err(out["cost"], ref["cost"]).max() < 1e-9
            ctx.close()
        finally:
            os.environ.pop("MP_SPLIT_CLEARANCE", None)


@pytest.mark.parametrize("variant", ["axis_aligned", "rotated", "tilted", "many_groups"])
def test_deep_and_tied_penetrations_match_oracle(variant, oracle_lib):
    """The culling bound of a penetrating sample (clearance_chunk: the mover's AABB shrunk by the depth found so far)
    must keep every box that can be as deep or deeper: nested boxes around the path (each deeper than the one before),
    exact duplicates (equal depths: the lowest index decides the argmin, hence the gradient) and boxes that only graze
    the mover.  Clearance profile and deciding-box gradient against the brute-force oracle."""
    from motion_planning_amd import _native as native
    from motion_planning_amd.blocks import Block
    from motion_planning_amd.synthetic import BoxList

    rng = np.random.default_rng({"axis_aligned": 1, "rotated": 2, "tilted": 3, "many_groups": 4}[variant])
    start, goal = np.array([-1.5, -1.5, 1.2]), np.array([1.5, 1.5, 0.8])
    blocks = [Block(size=(4.0, 4.0, 0.1), position=(0.0, 0.0, 0.05), object_id="table")]
    mid = 0.5 * (start + goal)
    for k in range(6):            # nested around the middle of the path, outermost first
        sz = 2.4 - 0.35 * k
        blocks.append(Block(size=(sz, sz, sz * 0.8), position=tuple(mid + 0.02 * k), object_id=f"nest_{k}"))
    for k in range(4):            # exact duplicates
        blocks.append(Block(size=(0.8, 0.7, 0.9), position=(-0.6, -0.7, 1.0), object_id=f"dup_{k}"))
    n_more = 300 if variant == "many_groups" else 20
    for k in range(n_more):       # clutter that overlaps the path corridor
        t = rng.random()
        c = start + t * (goal - start) + rng.normal(size=3) * 0.35
        blocks.append(Block(size=tuple(rng.uniform(0.1, 0.9, 3)), position=tuple(c), object_id=f"c_{k}"))
    prob = build_path_problem(BoxList(blocks), start, goal, native.build_basis, n_vias=2, n_samples_curve=101,
                              moving_block_size=(0.9, 0.6, 0.6), start_yaw_deg=0.0, goal_yaw_deg=90.0, combined_4d=True,
                              n_yaw_vias=2, safety_margin=0.02, preferred_safety_margin=0.08, w_safe=80.0,
                              w_safe_preferred=5.0, w_approach_collision=50.0, w_approach_clearance=5.0,
                              approach_only_clearance=0.05)
    if variant == "rotated":
        prob = H.with_rotated_statics(prob, seed=5)
    elif variant == "tilted":
        prob = H.with_tilted_statics(prob, seed=6, fraction=0.5)
    X = prob.x0 + 0.4 * prob.sigma0 * rng.normal(size=(96, prob.dim))
    ref = oracle_lib.evaluate(prob, X)
    assert (ref["dist"] < -0.2).mean() > 0.2, "the scene is meant to produce deep penetrations"
    for force in ("0", "1"):
        import os

        os.environ["MP_SPLIT_CLEARANCE"] = force
        try:
            ctx = native.Context(0)
            ctx.set_problem(prob)
            out = ctx.eval(X, want_grad=True, want_dist=True)
            _compare(out, ref, np.float64)
            ctx.close()
        finally:
            os.environ.pop("MP_SPLIT_CLEARANCE", None)
