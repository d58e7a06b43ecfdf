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
