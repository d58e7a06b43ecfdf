"""Scene bookkeeping and the YAML world model (host logic, no GPU)."""
import json

import numpy as np
import pytest

from motion_planning_amd.scene import Scene
from motion_planning_amd.scenarios import WorldModel, scenario_from_resolved
from tests import helpers as H

YAML_TEXT = """
defaults:
  base_size: [0.6, 0.9, 0.6]
scenarios:
  face_case:
    blocks:
      - {id: table, size: [2.0, 2.0, 0.1], position: [0.0, 0.0, 0.05]}
      - {id: cube, position: [0.0, -0.5, 0.4]}
    moving_block:
      start: [-0.4, -0.2, 1.5]
      goal_yaw_deg: 90.0
      goal: {type: face, base: cube, face: front, gap: 0.0, size: [0.9, 0.6, 0.6]}
      goal_normals: [[2.0, 0.0, 0.0]]
  between_case:
    blocks:
      - {id: a, position: [-0.6, 0.0, 0.35]}
      - {id: b, position: [0.6, 0.2, 0.35]}
    moving_block:
      size: [0.3, 0.3, 0.3]
      start: [0, 0, 1]
      goal: {type: between, ids: [a, b], position: [null, null, 0.5]}
      goal_normals: [[1, 0, 0], [-1, 0, 0]]
  point_case:
    moving_block:
      start: [0, 0, 1]
      goal: {type: point, position: [1, 2, 3]}
      goal_normals: [[0, 0, 3]]
"""


def test_world_model_goal_types(tmp_path):
    f = tmp_path / "s.yaml"
    f.write_text(YAML_TEXT)
    wm = WorldModel(f)
    assert wm.list_scenarios() == ["between_case", "face_case", "point_case"]
    sc = wm.build_scenario("FACE_case ")
    assert sc.goal == pytest.approx((0.0, -0.5 + 0.45 + 0.3, 0.4))
    assert sc.moving_block_size == (0.6, 0.9, 0.6) and sc.goal_yaw_deg == 90.0 and sc.start_yaw_deg == 0.0
    assert sc.goal_normals == ((1.0, 0.0, 0.0),)
    assert [b.object_id for b in sc.scene.blocks] == ["table", "cube"]
    sc = wm.build_scenario("between_case")
    assert sc.goal == pytest.approx((0.0, 0.1, 0.5))
    assert wm.build_scenario("point_case").goal == (1.0, 2.0, 3.0)
    with pytest.raises(ValueError, match="Unknown scenario"):
        wm.build_scenario("nope")


def test_world_model_errors(tmp_path):
    f = tmp_path / "bad.yaml"
    f.write_text("foo: 1\n")
    with pytest.raises(ValueError, match="Invalid scenarios YAML"):
        WorldModel(f)
    f.write_text("scenarios:\n  a:\n    moving_block:\n      start: [0,0,0]\n      goal: {type: point, position: [0,0,1]}\n")
    with pytest.raises(ValueError, match="missing moving_block.goal_normals"):
        WorldModel(f).build_scenario("a")
    f.write_text("scenarios:\n  a:\n    moving_block:\n      start: [0,0,0]\n      goal: {type: warp}\n      goal_normals: [[0,0,1]]\n")
    with pytest.raises(ValueError, match="Unknown goal type"):
        WorldModel(f).build_scenario("a")
    f.write_text("scenarios:\n  a:\n    moving_block:\n      start: [0,0,0]\n      goal: {type: point, position: [0,0,1]}\n      goal_normals: [[0,0,0]]\n")
    with pytest.raises(ValueError, match="must be non-zero"):
        WorldModel(f).build_scenario("a")


def test_resolved_fixture_matches_reference_scenarios():
    sc = H.load_scenarios()
    assert sorted(sc) == ["between", "front", "on_top", "step_01_first_on_ground", "step_02_second_in_front",
                          "step_03_third_on_top", "step_04_between_two_blocks"]
    assert sc["front"]["goal"] == pytest.approx([0.0, 0.25, 0.4])
    assert sc["on_top"]["goal"] == pytest.approx([0.05, -0.2, 0.95])
    assert sc["step_02_second_in_front"]["goal"] == pytest.approx([0.0, 0.6, 0.4])
    cfg = scenario_from_resolved(sc["between"])
    assert len(cfg.scene.blocks) == 4 and cfg.goal == pytest.approx((0.0, 0.0, 0.35)) and cfg.start_yaw_deg == -15.0


def test_scene_ids_and_stacking():
    s = Scene()
    assert s.add_block((1, 1, 1), (0, 0, 0.5)) == "obj_0"
    assert s.add_block((1, 2, 1), (3, 0, 0.5), object_id="b") == "b"
    with pytest.raises(ValueError, match="already exists"):
        s.add_block((1, 1, 1), (0, 0, 0), object_id="b")
    with pytest.raises(KeyError):
        s.get_block("zz")
    with pytest.raises(IndexError):
        s.get_block(5)
    assert s.get_block(1) is s.get_block("b")
    np.testing.assert_allclose(s.get_top_point("b", (0.2, 0.2, 0.4), gap=0.1, xy_offset=(0.1, -0.1)), [3.1, -0.1, 1.3])
    np.testing.assert_allclose(s.get_bottom_point("b", (0.2, 0.2, 0.4)), [3, 0, -0.2])
    np.testing.assert_allclose(s.get_front_point("b", (0.2, 0.2, 0.4)), [3, 1.1, 0.5])
    np.testing.assert_allclose(s.get_back_point("b", (0.2, 0.2, 0.4), xz_offset=(0.5, 0.25)), [3.5, -1.1, 0.75])
    np.testing.assert_allclose(s.get_right_point("b", (0.2, 0.2, 0.4)), [3.6, 0, 0.5])
    np.testing.assert_allclose(s.get_left_point("b", (0.2, 0.2, 0.4), yz_offset=(0.3, 0.0)), [2.4, 0.3, 0.5])
    with pytest.raises(ValueError, match="face must be one of"):
        s.get_stack_point_on_face("b", (1, 1, 1), "side")
    # rotated base: 90 deg yaw swaps the local axes
    q = (0.0, 0.0, np.sin(np.pi / 4), np.cos(np.pi / 4))
    s.add_block((1, 2, 1), (0, 5, 0.5), quat=q, object_id="r")
    np.testing.assert_allclose(s.get_right_point("r", (0.2, 0.2, 0.2)), [0, 5.6, 0.5], atol=1e-12)
    oid = s.stack_on("b", (0.5, 0.5, 0.5), object_id="top")
    assert oid == "top" and s.get_block("top").position == pytest.approx((3, 0, 1.25))
    s.stack_on_face("b", (0.5, 0.5, 0.5), "left", object_id="l")
    assert s.get_block("l").position == pytest.approx((2.25, 0, 0.5))
    blk = s.get_block("b")
    assert blk.contains((3.5, 1.0, 1.0)) and not blk.contains((3.51, 0, 0.5))
    assert blk.vertices_world().shape == (8, 3)


def test_step_scenario_generator_reproduces_the_reference_data_file():
    """generate_step_scenarios() == the parsed example/generated_scenarios.yaml (golden: generated_payload.json),
    and the placement helper follows the per-direction conventions of geom/planning.py:47-87."""
    import json
    from pathlib import Path

    from motion_planning_amd.blocks import Block
    from motion_planning_amd.planning import goal_and_via_for_placement
    from motion_planning_amd.synthetic import generate_step_scenarios

    gold = json.loads((Path(__file__).parent / "golden" / "generated_payload.json").read_text())
    assert generate_step_scenarios() == gold

    class One:
        def __init__(self, b):
            self.b = b

        def get_block(self, _):
            return self.b

    blk = One(Block((0.9, 0.6, 0.4), (1.0, 2.0, 3.0)))
    kw = dict(gap=0.01, lateral_offset=(0.1, 0.2), interlock_x=0.05)
    # values derived by hand from the reference's formulas (the module itself needs python-fcl to import)
    want = {"front": ((1.1, 2.61, 3.2), (1.1, 2.66, 3.45)), "back": ((1.1, 2.0, 2.29), (1.1, 2.0, 2.49)),
            "left": ((1.91, 2.1, 3.2), (1.96, 2.1, 3.45)), "right": ((0.09, 2.1, 3.2), (0.04, 2.1, 3.45)),
            "top": ((1.15, 2.2, 3.21), (1.15, 2.2, 3.51))}
    for d, (g, v) in want.items():
        goal, via = goal_and_via_for_placement(blk, 0, d.upper() if d == "top" else d, **kw)
        np.testing.assert_allclose(goal, g, atol=1e-12)
        np.testing.assert_allclose(via, v, atol=1e-12)
    # a block turned by 90 deg about z: local x -> world y
    h = np.sqrt(0.5)
    turned = One(Block((0.9, 0.6, 0.4), (0.0, 0.0, 0.0), quat=(0.0, 0.0, h, h)))
    goal, via = goal_and_via_for_placement(turned, 0, "left", gap=0.0, via_shift_along_normal=0.0, via_offset_local=(0.1, 0.0, 0.0))
    np.testing.assert_allclose(goal, (0.0, 0.9, 0.0), atol=1e-12)
    np.testing.assert_allclose(via, (0.0, 1.0, 0.0), atol=1e-12)
    with pytest.raises(ValueError):
        goal_and_via_for_placement(blk, 0, "under")


def test_placement_helper_equals_the_reference():
    """tests/golden/g6_placement.json: 28 placements computed by the reference's own `goal_and_via_for_placement`
    (geom/planning.py:6-95, run from its function body by tools/gen_golden.py with the reference's quat_to_rot): all five
    directions, mixed case, identity / yaw-only / general quaternions, default and random keyword arguments."""
    import json

    import numpy as np
    import pytest

    from motion_planning_amd.planning import goal_and_via_for_placement
    from tests import helpers as H

    g6 = json.loads((H.GOLDEN / "g6_placement.json").read_text())
    assert len(g6["cases"]) == 28 and {c["direction"].lower() for c in g6["cases"]} == {"front", "back", "left", "right", "top"}
    for c in g6["cases"]:
        s = Scene()
        s.add_block(tuple(c["size"]), tuple(c["position"]), quat=tuple(c["quat"]), object_id="b")
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c["kwargs"].items()}
        goal, via = goal_and_via_for_placement(s, "b", c["direction"], **kw)
        np.testing.assert_allclose(goal, c["goal"], rtol=0, atol=2e-15)     # same products, summed in the reference's order
        np.testing.assert_allclose(via, c["via"], rtol=0, atol=2e-15)
    with pytest.raises(ValueError) as ei:
        goal_and_via_for_placement(s, "b", "bottom")
    assert str(ei.value) == g6["bad_direction_message"]
