"""The CPU oracle against the golden vectors captured from the real reference module
(tests/golden/, generator tools/gen_golden.py).  Tolerance: 1e-9 relative (north_star)."""
import numpy as np
import pytest

from motion_planning_amd import _abi as abi
from motion_planning_amd import _native
from tests import helpers as H

CASES = H.load_cases()
G3 = H.load_g3()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_objective_terms_match_reference(case, oracle_lib):
    prob = H.problem_for_case(case)
    name = case["name"]
    X, ref = G3[f"{name}__X"], G3[f"{name}__out"]
    np.testing.assert_array_equal(prob.x0, G3[f"{name}__x0"])
    np.testing.assert_array_equal(prob.sigma0, G3[f"{name}__sigma0"])
    out = oracle_lib.evaluate(prob, X)
    # clearance profile the reference saw (same oracle geometry, reference-side trig / SciPy spline)
    dref = G3[f"{name}__dist"]
    for i in range(X.shape[0]):
        n_ref = int(np.sum(~np.isnan(dref[i])))
        assert n_ref in (prob.n_safe, prob.n_safe - prob.last_if_not_close)
        np.testing.assert_allclose(out["dist"][i, :n_ref], dref[i, :n_ref], rtol=1e-9, atol=1e-12)
    assert H.rel_err(out["cost"], ref[:, 0]).max() < 1e-9
    err = np.abs(out["terms"] - ref[:, 1:]) / np.maximum(np.abs(ref[:, 1:]), 1e-9)
    assert err.max() < 1e-9, (name, err.max(axis=0))


def test_spline_sampling_matches_scipy_path():
    g1 = np.load(H.GOLDEN / "g1_spline.npz")
    for key in g1.files:
        if not key.startswith("W_"):
            continue
        _, C, S = key.split("_")
        C, S = int(C), int(S)
        M = _native.build_basis(C, 3, S)
        np.testing.assert_allclose(M @ g1[key], g1[f"Q_{C}_{S}"], rtol=0, atol=2e-13)
        np.testing.assert_allclose(M.sum(axis=1), 1.0, atol=1e-14)
    for n in (2, 3, 4, 5):
        M = _native.build_basis(n, min(3, n - 1), 77)
        np.testing.assert_allclose(M @ g1[f"y_{n}"], g1[f"yq_{n}"], rtol=0, atol=2e-13)


def test_trig_matches_reference_quaternion(oracle_lib):
    g2 = np.load(H.GOLDEN / "g2_terms.npz")
    yaw, q = g2["quat_yaw"], g2["quat"]
    c_ref = 1 - 2 * (q[:, 2] * q[:, 2])
    s_ref = 2 * (q[:, 3] * q[:, 2])
    c, s = oracle_lib.yaw_rot(yaw)
    np.testing.assert_allclose(c, c_ref, rtol=0, atol=1e-15)
    np.testing.assert_allclose(s, s_ref, rtol=0, atol=1e-15)
    assert oracle_lib.yaw_rot(np.array([0.0]))[0][0] == 1.0 and oracle_lib.yaw_rot(np.array([0.0]))[1][0] == 0.0
    c32, s32 = oracle_lib.yaw_rot(yaw, dtype=np.float32)
    np.testing.assert_allclose(c32, c_ref, atol=3e-6)
    np.testing.assert_allclose(s32, s_ref, atol=3e-6)


def test_via_initialisation_matches_reference():
    from motion_planning_amd.problem import default_via_initialization

    g2 = np.load(H.GOLDEN / "g2_terms.npz")
    for n in (0, 1, 2, 5):
        got = default_via_initialization(np.array([-1.0, 0.5, 2.0]), np.array([2.0, -1.0, 0.3]), n)
        np.testing.assert_array_equal(got, g2[f"via_init_{n}"])


def test_host_helpers_match_reference():
    from motion_planning_amd import spline_opt as so

    g2 = np.load(H.GOLDEN / "g2_terms.npz")
    for S in (2, 3, 5, 64, 101):
        P, ref = g2[f"P_{S}"], g2[f"out_{S}"]
        assert so.path_length(P) == pytest.approx(ref[0], rel=1e-14)
        assert so.mean_turn_angle_deg(P) == pytest.approx(ref[3], rel=1e-12, abs=1e-12)
        # the other public helpers of the reference module (ref :63-79, :100-106, :117-145), same goldens
        assert so.curvature_cost(P) == pytest.approx(ref[1], rel=1e-12, abs=1e-300)
        assert so.yaw_smoothness_cost(g2[f"yaw_{S}"]) == pytest.approx(ref[2], rel=1e-13, abs=1e-300)
        assert so._goal_approach_alignment_cost(P, g2[f"normals_{S}"], 0.12) == pytest.approx(ref[4], rel=1e-12, abs=1e-15)
        assert so._goal_approach_alignment_cost(P, np.zeros((1, 3)), 0.12) == ref[5] == 0.0
    np.testing.assert_array_equal(so._normalize([3.0, 0.0, 4.0]), [0.6, 0.0, 0.8])
    np.testing.assert_array_equal(so._normalize(np.zeros(3)), np.zeros(3))
    for y, q in zip(g2["quat_yaw"], g2["quat"]):
        assert so.yaw_deg_to_quat(y) == tuple(q)


def test_window_integers_of_the_demo_config():
    # S=101 -> n_tail=26, keep_n=75, tail_n=13, 92 approach samples (strict <)
    case = next(c for c in CASES if c["name"] == "base_front")
    p = H.problem_for_case(case)
    assert (p.n_safe, p.keep_n, p.tail_start, p.goal_tail_n) == (101, 75, 101 - 26, 13)
    assert int(p.approach_mask.sum()) == 92 and p.approach_mask[91] == 1 and p.approach_mask[92] == 0
    case = next(c for c in CASES if c["name"] == "synth50_seed0")
    p = H.problem_for_case(case)
    assert (p.n_safe, p.keep_n, p.tail_start, p.goal_tail_n, int(p.approach_mask.sum())) == (64, 48, 48, 8, 58)
    case = next(c for c in CASES if c["name"] == "front_sub3")
    p = H.problem_for_case(case)
    assert p.idx_safe[-1] == 100 and p.idx_safe[-2] == 99 and p.n_safe == 35


def test_reference_shaped_numpy_path_matches_the_reference_goldens(oracle_lib):
    """oracle/numpy_path.py (SciPy spline per call, NumPy terms, per-sample distance loop) against the 16-tuples the
    real reference returned (G3), and against the C oracle's clearance profile."""
    from oracle import numpy_path as npth
    from tests import helpers as H

    g3 = H.load_g3()
    for case in H.load_cases():
        prob = H.problem_for_case(case)
        name = case["name"]
        X, gold = g3[f"{name}__X"], g3[f"{name}__out"]
        ref = oracle_lib.evaluate(prob, X)
        for i in range(0, X.shape[0], 3):
            cost, terms, d = npth.objective_single(prob, X[i])
            assert cost == pytest.approx(gold[i, 0], rel=1e-10, abs=1e-12), (name, i)
            np.testing.assert_allclose(terms, gold[i, 1:], rtol=1e-9, atol=1e-12, err_msg=f"{name}[{i}]")
            np.testing.assert_allclose(d, ref["dist"][i], rtol=1e-12, atol=1e-13, err_msg=f"{name}[{i}] clearance")


def test_path_distance_helpers_with_a_duck_typed_scene(oracle_lib):
    """_path_distances / safety_cost (ref :148-211) against any scene object with the reference's two methods: the three
    modes, the shape check of the per-sample quaternions and the squared-deficit sum."""
    from motion_planning_amd import spline_opt as so

    sc = H.load_scenarios()["front"]
    scene = oracle_lib.OracleScene(H.blocks_from(sc["blocks"]))
    P = np.linspace([-0.4, -0.2, 1.5], [0.0, 0.25, 0.45], 9)
    d_pt = so._path_distances(scene, P)
    np.testing.assert_array_equal(d_pt, [scene.signed_distance(p) for p in P])
    yaw = np.linspace(0.0, 90.0, 9)
    quats = np.array([so.yaw_deg_to_quat(y) for y in yaw])
    d_box = so._path_distances(scene, P, moving_block_size=(0.6, 0.9, 0.6), moving_block_quats=quats, ignore_ids=["table"])
    np.testing.assert_array_equal(d_box, [scene.signed_distance_block((0.6, 0.9, 0.6), p, tuple(q), ["table"]) for p, q in zip(P, quats)])
    d_fix = so._path_distances(scene, P, moving_block_size=(0.6, 0.9, 0.6))
    np.testing.assert_array_equal(d_fix, [scene.signed_distance_block((0.6, 0.9, 0.6), p) for p in P])
    with pytest.raises(ValueError, match=r"moving_block_quats must have shape \(len\(P\), 4\)"):
        so._path_distances(scene, P, moving_block_size=(0.6, 0.9, 0.6), moving_block_quats=quats[:-1])
    want = float(np.sum(np.maximum(0.0, 0.3 - d_box) ** 2))
    assert so.safety_cost(scene, P, 0.3, moving_block_size=(0.6, 0.9, 0.6), moving_block_quats=quats, ignore_ids=["table"]) == pytest.approx(want, rel=1e-15)
