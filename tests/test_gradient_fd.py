"""The analytic (sub)gradient has no reference counterpart (the reference is derivative-free,
SURVEY.md §0 F1): it is checked against central finite differences of the oracle's own fp64
cost, at points away from the kinks of max/min/clip."""
import numpy as np
import pytest

from tests import helpers as H

CASES = {c["name"]: c for c in H.load_cases()}


def _fd(orc, prob, x, h):
    D = x.size
    Xp = np.repeat(x[None, :], 2 * D, axis=0)
    for i in range(D):
        Xp[2 * i, i] += h
        Xp[2 * i + 1, i] -= h
    c = orc.evaluate(prob, Xp, want_grad=False, want_dist=False)["cost"]
    return (c[0::2] - c[1::2]) / (2 * h)


@pytest.mark.parametrize("name", ["base_front", "base_between", "synth50_seed0", "front_sep_yaw1", "front_point",
                                  "front_sub3", "front_vias3_s121", "vias_point", "vias_box_sub5",
                                  "base_step_04_between_two_blocks"])
def test_gradient_matches_central_differences(name, oracle_lib):
    prob = H.problem_for_case(CASES[name])
    X = H.load_g3()[f"{name}__X"]
    out = oracle_lib.evaluate(prob, X)
    good = 0
    # candidate 0 is x0, the straight line: in the symmetric scenarios it sits exactly on a kink
    # (|dx| at dx = 0, tied SAT axes) where a symmetric stencil cancels; skip it
    for i in range(1, X.shape[0]):
        g = out["grad"][i]
        f1, f2 = _fd(oracle_lib, prob, X[i], 1e-6), _fd(oracle_lib, prob, X[i], 2e-6)
        scale = np.abs(g).max() + 1e-9
        if np.abs(f1 - f2).max() > 1e-4 * scale:   # the two step sizes disagree: a kink is inside the stencil
            continue
        assert np.abs(g - f1).max() <= 2e-5 * scale, (name, i, g, f1)
        good += 1
    assert good >= (X.shape[0] - 1) // 2, f"{name}: too few smooth points ({good}/{X.shape[0]})"


@pytest.mark.parametrize("name", ["base_between", "synth50_seed0", "front_point", "vias_box_sub5"])
def test_gradient_with_rotated_static_boxes(name, oracle_lib):
    """Static boxes with a yaw of their own: the pair gradient is computed in the box frame and rotated back."""
    prob = H.with_rotated_statics(H.problem_for_case(CASES[name]), seed=3)
    assert prob.boxes.shape[1] == 8
    X = H.load_g3()[f"{name}__X"]
    out = oracle_lib.evaluate(prob, X)
    base = oracle_lib.evaluate(H.problem_for_case(CASES[name]), X)
    assert not np.array_equal(out["dist"], base["dist"])     # the rotation really changed the scene
    good = 0
    for i in range(1, X.shape[0]):
        g = out["grad"][i]
        f1, f2 = _fd(oracle_lib, prob, X[i], 1e-6), _fd(oracle_lib, prob, X[i], 2e-6)
        scale = np.abs(g).max() + 1e-9
        if np.abs(f1 - f2).max() > 1e-4 * scale:
            continue
        assert np.abs(g - f1).max() <= 2e-5 * scale, (name, i, g, f1)
        good += 1
    assert good >= (X.shape[0] - 1) // 3, f"{name}: too few smooth points ({good}/{X.shape[0]})"


@pytest.mark.parametrize("name", ["base_between", "synth50_seed0", "front_point"])
def test_gradient_with_tilted_static_boxes(name, oracle_lib):
    """Static boxes with a general rotation: pair gradients by central differences of the exact pair value (box model),
    closed form through the box frame (point model)."""
    prob = H.with_tilted_statics(H.problem_for_case(CASES[name]), seed=4)
    assert prob.boxes.shape[1] == 10
    X = H.load_g3()[f"{name}__X"]
    out = oracle_lib.evaluate(prob, X)
    good = 0
    for i in range(1, X.shape[0]):
        g = out["grad"][i]
        f1, f2 = _fd(oracle_lib, prob, X[i], 1e-6), _fd(oracle_lib, prob, X[i], 2e-6)
        scale = np.abs(g).max() + 1e-9
        if np.abs(f1 - f2).max() > 1e-4 * scale:
            continue
        assert np.abs(g - f1).max() <= 5e-5 * scale, (name, i, g, f1)
        good += 1
    assert good >= (X.shape[0] - 1) // 3, f"{name}: too few smooth points ({good}/{X.shape[0]})"
