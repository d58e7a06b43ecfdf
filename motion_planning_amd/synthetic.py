"""Synthetic scenes and candidate batches for the throughput configurations of BASELINE.json
(SURVEY.md §8d): config 3 = 65,536 candidates x 64 samples x 50 boxes; config 4 = 2,000 boxes,
256 samples.  Shapes and constants follow the reference's scenario generator
(example/generate_scenarios.py: table 4x4x0.1, moving block 0.9x0.6x0.6)."""
from __future__ import annotations

from typing import Tuple

import numpy as np

from .blocks import Block
from .configs import BASE_CONFIG
from .problem import PathProblem, build_path_problem


class BoxList:
    """Minimal duck-typed scene: just ``blocks`` (enough for problem.scene_to_boxes)."""

    def __init__(self, blocks):
        self.blocks = list(blocks)


def random_scene(n_boxes: int, seed: int, size_range: Tuple[float, float], start, goal) -> BoxList:
    """Table + (n_boxes - 1) random axis-aligned boxes in [-2,2]^2 x [0,2] that stay clear
    of the moving block's start and goal poses (hence never contain start or goal)."""
    rng = np.random.default_rng(seed)
    blocks = [Block(size=(4.0, 4.0, 0.1), position=(0.0, 0.0, 0.05), object_id="table")]
    start = np.asarray(start, float)
    goal = np.asarray(goal, float)
    lo = np.array([-2.0, -2.0, 0.0])
    hi = np.array([2.0, 2.0, 2.0])
    while len(blocks) < n_boxes:
        centre = lo + (hi - lo) * rng.random(3)
        size = size_range[0] + (size_range[1] - size_range[0]) * rng.random(3)
        h = 0.5 * size
        # keep the start and goal poses of the moving block free at any yaw (and so the boxes
        # never contain start/goal): half diagonal of the 0.9 x 0.6 footprint is 0.54
        keep = np.array([0.75, 0.75, 0.45])
        if np.all(np.abs(start - centre) <= h + keep) or np.all(np.abs(goal - centre) <= h + keep):
            continue
        blocks.append(Block(size=tuple(size.tolist()), position=tuple(centre.tolist()), object_id=f"box_{len(blocks)}"))
    return BoxList(blocks)


def synthetic_problem(basis_fn, n_boxes: int = 50, n_samples: int = 64, seed: int = 0,
                      size_range: Tuple[float, float] = (0.1, 0.6)) -> PathProblem:
    """Config-3-shaped problem: yaw 0 -> 90 deg across a cluttered workspace, BASE_CONFIG weights."""
    start = (-1.5, -1.5, 1.5)
    goal = (1.5, 1.5, 0.4)
    scene = random_scene(n_boxes, seed, size_range, start, goal)
    cfg = dict(BASE_CONFIG)
    cfg["n_samples_curve"] = int(n_samples)
    return build_path_problem(
        scene, start, goal, basis_fn, moving_block_size=(0.9, 0.6, 0.6), start_yaw_deg=0.0, goal_yaw_deg=90.0,
        goal_approach_normals=np.array([[0.0, 0.0, 1.0]]), **cfg,
    )


def synthetic_candidates(problem: PathProblem, n: int, seed: int = 0, dtype=np.float64) -> np.ndarray:
    """Multi-start candidates x = x0 + sigma0 * eps, eps ~ N(0, 1) (the CEM's first generation,
    geom/spline_opt.py:249-250)."""
    rng = np.random.default_rng(seed)
    eps = rng.normal(size=(n, problem.dim))
    return np.ascontiguousarray(problem.x0[None, :] + eps * problem.sigma0[None, :], dtype=dtype)


# ---- scenario-parallel workload (BASELINE config 5) -------------------------------------------
# Randomised wall-building steps in the spirit of the reference's example/generate_scenarios.py:
# 0.9 x 0.6 x 0.6 blocks on a block-size-aligned grid over the 4 x 4 x 0.1 table, the moving block
# is to be placed against already placed blocks (on the ground, beside, on top, in a gap), goal
# normals come from the faces it will touch, and the start follows the reference's rule
# (goal - 1.2 * approach direction, lifted to max(goal_z + 0.9, 1.25); generate_scenarios.py:144-150).
_BLOCK = (0.9, 0.6, 0.6)
_TABLE_TOP = 0.1


def generate_wall_scenario(rng: np.random.Generator):
    """One random placement step: returns (blocks, start, goal, goal_normals)."""
    gx, gy, gz = _BLOCK
    n_placed = int(rng.integers(1, 13))
    occupied = {}
    # grow a small wall: new blocks go on the ground next to the wall or on top of a supported cell
    ci, cj = int(rng.integers(-1, 2)), int(rng.integers(-2, 3))
    occupied[(ci, cj, 0)] = True
    guard = 0
    while len(occupied) < n_placed and guard < 200:
        guard += 1
        i, j, k = list(occupied)[int(rng.integers(0, len(occupied)))]
        move = int(rng.integers(0, 5))
        cand = [(i + 1, j, k), (i - 1, j, k), (i, j + 1, k), (i, j - 1, k), (i, j, k + 1)][move]
        if cand in occupied or abs(cand[0]) > 1 or abs(cand[1]) > 2 or cand[2] > 2:
            continue
        if cand[2] > 0 and (cand[0], cand[1], cand[2] - 1) not in occupied:
            continue
        occupied[cand] = True
    # goal: a free, supported cell that touches the wall
    options = []
    for (i, j, k) in occupied:
        for cand in [(i + 1, j, k), (i - 1, j, k), (i, j + 1, k), (i, j - 1, k), (i, j, k + 1)]:
            if cand in occupied or abs(cand[0]) > 1 or abs(cand[1]) > 2 or cand[2] > 2:
                continue
            if cand[2] > 0 and (cand[0], cand[1], cand[2] - 1) not in occupied:
                continue
            if (cand[0], cand[1], cand[2] + 1) in occupied:
                continue  # must be reachable from above
            options.append(cand)
    if not options:
        options = [(ci, cj, max(k for (i, j, k) in occupied if (i, j) == (ci, cj)) + 1)]
    gi, gj, gk = options[int(rng.integers(0, len(options)))]

    def centre(i, j, k):
        return (i * gx, j * gy, _TABLE_TOP + 0.5 * gz + k * gz)

    goal = centre(gi, gj, gk)
    normals = []
    for (di, dj, n) in [(-1, 0, (1.0, 0.0, 0.0)), (1, 0, (-1.0, 0.0, 0.0)), (0, -1, (0.0, 1.0, 0.0)), (0, 1, (0.0, -1.0, 0.0))]:
        if (gi + di, gj + dj, gk) in occupied:
            normals.append(n)
    normals.append((0.0, 0.0, 1.0))  # supported from below (table or block)
    s = np.sum(np.asarray(normals), axis=0)
    nrm = float(np.linalg.norm(s))
    approach = -s / nrm if nrm > 1e-12 else np.array([0.0, 0.0, -1.0])
    start = (goal[0] - 1.2 * approach[0], goal[1] - 1.2 * approach[1], max(goal[2] + 0.9, 1.25))
    blocks = [Block(size=(4.0, 4.0, 0.1), position=(0.0, 0.0, 0.05), object_id="table")]
    for n, (i, j, k) in enumerate(sorted(occupied)):
        blocks.append(Block(size=_BLOCK, position=centre(i, j, k), object_id=f"block_{n + 1}"))
    return blocks, start, goal, np.asarray(normals, dtype=float)


def wall_scenario_problems(basis_fn, n_scenarios: int, seed: int = 2, n_samples: int = 101):
    """``n_scenarios`` PathProblems with BASE_CONFIG (one shared mp_config, per-scenario task data)."""
    rng = np.random.default_rng(seed)
    cache = {}

    def cached_basis(n, k, S):
        key = (n, k, S)
        if key not in cache:
            cache[key] = basis_fn(n, k, S)
        return cache[key]

    cfg = dict(BASE_CONFIG)
    cfg["n_samples_curve"] = int(n_samples)
    out = []
    for _ in range(n_scenarios):
        blocks, start, goal, normals = generate_wall_scenario(rng)
        out.append(build_path_problem(BoxList(blocks), start, goal, cached_basis, moving_block_size=_BLOCK,
                                      start_yaw_deg=0.0, goal_yaw_deg=0.0, goal_approach_normals=normals, **cfg))
    return out


# ---- the four deterministic construction steps (example/generate_scenarios.py:153-264) ------------------
def _r6(v: float) -> float:
    out = round(float(v), 6)
    return 0.0 if abs(out) < 1e-12 else out


def _supported_from_below(blocks, goal, moving_size) -> bool:
    """A block top lies directly under the moving block's bottom face at the goal (:94-117)."""
    bottom = goal[2] - 0.5 * moving_size[2]
    gaps = [abs(b["position"][2] + 0.5 * b["size"][2] - bottom) for b in blocks
            if abs(goal[0] - b["position"][0]) <= 0.5 * b["size"][0] + 1e-9
            and abs(goal[1] - b["position"][1]) <= 0.5 * b["size"][1] + 1e-9]
    return bool(gaps) and min(gaps) <= 1e-6


def _placement_step(blocks, goal_cfg, goal, contact_normals, fallback_approach=(0.0, -1.0, 0.0)):
    """moving_block entry of one step: contact normals (+ the support normal if it is new, :120-133),
    approach = minus their normalised sum (:74-85), start = goal - 1.2 approach lifted above the scene (:144-150)."""
    normals = [np.asarray(n, dtype=float) / max(np.linalg.norm(n), 1e-300) for n in contact_normals]
    up = np.array([0.0, 0.0, 1.0])
    if _supported_from_below(blocks, goal, _BLOCK) and all(abs(float(up @ n)) < 0.999 for n in normals):
        normals.append(up)
    total = np.sum(normals, axis=0)
    nrm = float(np.linalg.norm(total))
    if nrm < 1e-12:
        approach = np.asarray(fallback_approach, dtype=float)
        approach = approach / np.linalg.norm(approach)
    else:
        approach = -(total / nrm)
        approach = approach / np.linalg.norm(approach)
    start = (goal[0] - 1.2 * approach[0], goal[1] - 1.2 * approach[1], max(goal[2] + 0.9, 1.25))
    return {
        "size": list(_BLOCK),
        "start": [_r6(v) for v in start],
        "start_yaw_deg": 0.0,
        "goal_yaw_deg": 0.0,
        "goal": goal_cfg,
        "goal_normals": [[_r6(v) for v in n] for n in normals],
        "approach_direction": [_r6(v) for v in approach],
    }


def generate_step_scenarios() -> dict:
    """The payload of ``example/generated_scenarios.yaml``: first block on the ground, second in front of it,
    third on top, fourth between two blocks — all on the block-size grid."""
    gx, gy, gz = _BLOCK
    ground_z = _TABLE_TOP + 0.5 * gz
    table = {"id": "table", "size": [4.0, 4.0, 0.1], "position": [0.0, 0.0, 0.05]}
    blk = lambda oid, c: {"id": oid, "size": list(_BLOCK), "position": [_r6(v) for v in c]}
    face = lambda base, f: {"type": "face", "base": base, "face": f, "gap": 0.0, "tangential_offset": [0.0, 0.0], "size": list(_BLOCK)}
    first, front = (0.0, 0.0, ground_z), (gx, 0.0, ground_z)
    steps = {}
    b = [table]
    steps["step_01_first_on_ground"] = {"blocks": b, "moving_block": _placement_step(b, face("table", "top"), first, [(0.0, 0.0, 1.0)])}
    b = [table, blk("block_1", first)]
    steps["step_02_second_in_front"] = {"blocks": b, "moving_block": _placement_step(b, face("block_1", "front"), front, [(1.0, 0.0, 0.0)])}
    b = [table, blk("block_1", first), blk("block_2", front)]
    steps["step_03_third_on_top"] = {"blocks": b, "moving_block": _placement_step(b, face("block_2", "top"), (gx, 0.0, ground_z + gz), [(0.0, 0.0, 1.0)])}
    b = [table, blk("left_block", (-gx, 0.0, ground_z)), blk("right_block", (gx, 0.0, ground_z))]
    between = {"type": "between", "ids": ["left_block", "right_block"], "position": [None, None, _r6(ground_z)]}
    steps["step_04_between_two_blocks"] = {"blocks": b, "moving_block": _placement_step(b, between, first, [(1.0, 0.0, 0.0), (-1.0, 0.0, 0.0)])}
    return {"defaults": {"base_size": list(_BLOCK)}, "scenarios": steps}
