"""Scenario ingestion: YAML world model -> Scene + start/goal/normals.

Mirrors example/scenarios.py of the reference (WorldModel.list_scenarios / build_scenario,
the three goal types point/face/between at :94-129, unit-normalised goal normals at :132-161)
so that the reference's scenario files load unchanged.  Also reads the resolved-scenario JSON
fixtures under tests/golden/ (same ScenarioConfig, no YAML or stacking step needed).
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, List, Tuple, Union

import numpy as np

from .scene import Scene

Vec3 = Tuple[float, float, float]


@dataclass(frozen=True)
class ScenarioConfig:
    scene: Scene
    start: Vec3
    goal: Vec3
    moving_block_size: Vec3
    start_yaw_deg: float
    goal_yaw_deg: float
    goal_normals: Tuple[Vec3, ...]


def _vec(values, n: int = 3) -> Tuple[float, ...]:
    out = tuple(float(v) for v in values)
    if len(out) != n:
        raise ValueError(f"expected {n} values, got {len(out)}")
    return out


def _goal_from_face(scene: Scene, moving_size: Vec3, goal_cfg: Dict[str, Any]) -> Vec3:
    pos = scene.get_stack_point_on_face(
        base=goal_cfg["base"],
        new_size=_vec(goal_cfg.get("size", moving_size)),
        face=str(goal_cfg["face"]).lower(),
        gap=float(goal_cfg.get("gap", 0.0)),
        tangential_offset=tuple(float(v) for v in goal_cfg.get("tangential_offset", [0.0, 0.0])),
    )
    return _vec(pos.tolist())


def _goal_between(scene: Scene, goal_cfg: Dict[str, Any]) -> Vec3:
    a, b = goal_cfg["ids"][0], goal_cfg["ids"][1]
    mid = 0.5 * (np.asarray(scene.get_block(a).position, dtype=float) + np.asarray(scene.get_block(b).position, dtype=float))
    given = goal_cfg.get("position")
    if given is None:
        return _vec(mid.tolist())
    # a null component means "use the midpoint"
    return tuple(float(mid[i]) if given[i] is None else float(given[i]) for i in range(3))


def resolve_goal(scene: Scene, moving_size: Vec3, goal_cfg: Dict[str, Any]) -> Vec3:
    kind = str(goal_cfg.get("type", "point")).lower()
    if kind == "point":
        return _vec(goal_cfg["position"])
    if kind == "face":
        return _goal_from_face(scene, moving_size, goal_cfg)
    if kind == "between":
        return _goal_between(scene, goal_cfg)
    raise ValueError(f"Unknown goal type: {kind}")


def parse_goal_normals(moving_cfg: Dict[str, Any], scenario_name: str) -> Tuple[Vec3, ...]:
    raw = moving_cfg.get("goal_normals")
    if raw is None:
        raise ValueError(
            f"Scenario '{scenario_name}' is missing moving_block.goal_normals. Provide one or more 3D vectors in YAML."
        )
    if not isinstance(raw, list) or not raw:
        raise ValueError(f"Scenario '{scenario_name}' has invalid moving_block.goal_normals (must be a non-empty list).")
    out: List[Vec3] = []
    for i, item in enumerate(raw):
        if not isinstance(item, (list, tuple)) or len(item) != 3:
            raise ValueError(f"Scenario '{scenario_name}' goal_normals[{i}] is invalid; expected 3 values.")
        v = np.asarray([float(c) for c in item], dtype=float)
        mag = float(np.linalg.norm(v))
        if mag < 1e-12:
            raise ValueError(f"Scenario '{scenario_name}' goal_normals[{i}] must be non-zero.")
        u = v / mag
        out.append((float(u[0]), float(u[1]), float(u[2])))
    return tuple(out)


class WorldModel:
    """Scenario file reader with the reference's interface (example/scenarios.py:25-64)."""

    def __init__(self, scenarios_file: Union[Path, str]):
        self.scenarios_file = Path(scenarios_file)
        text = self.scenarios_file.read_text(encoding="utf-8")
        if self.scenarios_file.suffix.lower() == ".json":
            payload = json.loads(text)
        else:
            import yaml

            payload = yaml.safe_load(text)
        if not isinstance(payload, dict) or "scenarios" not in payload:
            raise ValueError(f"Invalid scenarios YAML: {self.scenarios_file}")
        if not isinstance(payload["scenarios"], dict):
            raise ValueError("'scenarios' must be a mapping")
        self.payload = payload

    def list_scenarios(self) -> List[str]:
        return sorted(self.payload["scenarios"].keys())

    def build_scenario(self, name: str) -> ScenarioConfig:
        table = self.payload["scenarios"]
        key = str(name).strip().lower()
        if key not in table:
            raise ValueError(f"Unknown scenario '{name}'. Available: {', '.join(sorted(table.keys()))}")
        base_size = _vec(self.payload.get("defaults", {}).get("base_size", [0.6, 0.9, 0.6]))
        cfg = table[key]
        scene = Scene()
        for item in cfg.get("blocks", []):
            scene.add_block(
                size=_vec(item.get("size", base_size)),
                position=_vec(item["position"]),
                quat=_vec(item.get("quat", [0.0, 0.0, 0.0, 1.0]), 4),
                object_id=str(item["id"]),
            )
        mv = cfg["moving_block"]
        moving_size = _vec(mv.get("size", base_size))
        return ScenarioConfig(
            scene=scene,
            start=_vec(mv["start"]),
            goal=resolve_goal(scene, moving_size, mv["goal"]),
            moving_block_size=moving_size,
            start_yaw_deg=float(mv.get("start_yaw_deg", 0.0)),
            goal_yaw_deg=float(mv.get("goal_yaw_deg", 0.0)),
            goal_normals=parse_goal_normals(mv, key),
        )


def list_scenarios(scenarios_file: Union[Path, str]) -> List[str]:
    return WorldModel(scenarios_file).list_scenarios()


def build_scenario(name: str, scenarios_file: Union[Path, str]) -> ScenarioConfig:
    return WorldModel(scenarios_file).build_scenario(name)


def scenario_from_resolved(entry: Dict[str, Any]) -> ScenarioConfig:
    """Build a ScenarioConfig from one entry of tests/golden/scenarios.json (already resolved)."""
    scene = Scene()
    for b in entry["blocks"]:
        scene.add_block(size=_vec(b["size"]), position=_vec(b["position"]), object_id=str(b["id"]))
    return ScenarioConfig(
        scene=scene, start=_vec(entry["start"]), goal=_vec(entry["goal"]),
        moving_block_size=_vec(entry["moving_block_size"]), start_yaw_deg=float(entry["start_yaw_deg"]),
        goal_yaw_deg=float(entry["goal_yaw_deg"]), goal_normals=tuple(_vec(n) for n in entry["goal_normals"]),
    )
