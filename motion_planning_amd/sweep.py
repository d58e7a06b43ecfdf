"""Planner benchmark / hyper-parameter sweep bookkeeping (counterpart of the reference's
example/benchmark_planners.py) on top of the native planner.

Reproduces, with the reference's own numbers: the random-search grids of
``_sample_strategy_config`` (ref :64-109), the scenario score (ref :112-117), the per-scenario
record and the aggregate keys (ref :120-200), the exception-to-sentinel rule (ref :203-236), the
train/test split (ref :343-348) and the per-method seed offsets (ref :365-367).  The RNG call
sequence is kept, so a sweep with the same ``--seed`` draws the same configurations.
"""
from __future__ import annotations

import json
import time
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np

from .configs import BASE_CONFIG
from .scenarios import ScenarioConfig
from .spline_opt import optimize_bspline_path

AGGREGATE_KEYS = (
    "score", "runtime_s", "fun", "length", "path_efficiency", "curvature_cost", "turn_angle_mean_deg",
    "yaw_smoothness_cost", "safety_cost", "preferred_safety_cost", "approach_rebound_cost", "goal_clearance_cost",
    "goal_clearance_target_cost", "approach_clearance_cost", "approach_collision_cost", "goal_approach_normal_cost",
    "min_clearance", "mean_clearance", "nit",
)
_INFO_FLOATS = AGGREGATE_KEYS[3:-1]
# wall clock of run_single (the reference's score has a 0.15 * runtime_s term, ref :112-117); tests pin it to make two
# sweeps comparable
_clock = time.perf_counter


def sample_strategy_config(method: str, rng: np.random.Generator) -> Tuple[Dict[str, Any], Dict[str, Any]]:
    """One random-search draw: planner weights + optimiser options for ``method``."""
    cfg = dict(BASE_CONFIG)
    for key, grid in (("init_offset_scale", [0.5, 0.7, 1.0]), ("w_len", [3.5, 5.0, 6.0]), ("w_curv", [0.08, 0.12, 0.18]),
                      ("w_safe", [260.0, 380.0, 520.0]), ("w_goal_approach_normal", [40.0, 80.0, 120.0]),
                      ("w_approach_collision", [1000.0, 1400.0, 1800.0])):
        cfg[key] = float(rng.choice(grid))
    name = method.upper()

    def cem_block(iters):
        return {
            "population_size": int(rng.choice([48, 64, 96])),
            "elite_frac": float(rng.choice([0.15, 0.2, 0.25])),
            "max_iter": int(rng.choice(iters)),
            "alpha": float(rng.choice([0.6, 0.7, 0.8])),
            "min_sigma": float(rng.choice([5e-4, 1e-3])),
            "tol": 1e-3,
            "seed": int(rng.integers(0, 1_000_000)),
        }

    def powell_block(iters):
        return {"maxiter": int(rng.choice(iters)), "xtol": float(rng.choice([3e-3, 1e-3])), "ftol": float(rng.choice([3e-3, 1e-3]))}

    if name == "POWELL":
        options = powell_block([80, 140, 220])
    elif name == "CEM":
        options = cem_block([60, 90, 140])
    elif name in {"CEM-POWELL", "HYBRID"}:
        options = {"cem": cem_block([45, 60, 90]), "powell": powell_block([60, 100, 140])}
    else:
        raise ValueError(f"Unsupported method: {method}")
    return cfg, options


def scenario_score(info: Dict[str, Any], runtime_s: float) -> float:
    """Lower is better; collisions and failed solves are penalised heavily (ref :112-117)."""
    min_clear = float(info.get("min_clearance", -1.0))
    penalty = 50_000.0 * max(0.0, -min_clear) ** 2
    if not bool(info.get("success", False)):
        penalty += 10_000.0
    return float(info["fun"]) + penalty + 0.15 * runtime_s


def run_single(sc: ScenarioConfig, name: str, method: str, config: Dict[str, Any], options: Dict[str, Any],
               planner: Callable = optimize_bspline_path, **planner_extra) -> Dict[str, Any]:
    t0 = _clock()
    _, _, info = planner(
        scene=sc.scene, start=sc.start, goal=sc.goal, moving_block_size=sc.moving_block_size,
        start_yaw_deg=sc.start_yaw_deg, goal_yaw_deg=sc.goal_yaw_deg,
        goal_approach_normals=np.asarray(sc.goal_normals, dtype=float), method=method, options=options, **config,
        **planner_extra,
    )
    dt = float(_clock() - t0)
    straight = float(np.linalg.norm(np.asarray(sc.goal, dtype=float) - np.asarray(sc.start, dtype=float)))
    rec = {"scenario": name, "runtime_s": dt, "score": scenario_score(info, dt), "success": bool(info.get("success", False)),
           "fun": float(info.get("fun", 1e9))}
    for key in _INFO_FLOATS:
        if key == "path_efficiency":
            rec[key] = float(info.get("length", 0.0)) / max(straight, 1e-9)
        else:
            rec[key] = float(info.get(key, -1.0 if key == "min_clearance" else 0.0))
    rec["nit"] = int(info.get("nit", 0))
    rec["message"] = str(info.get("message", ""))
    return rec


def aggregate_numeric(per_scenario: List[Dict[str, Any]]) -> Dict[str, Dict[str, float]]:
    out: Dict[str, Dict[str, float]] = {}
    for key in AGGREGATE_KEYS:
        vals = np.asarray([float(r[key]) for r in per_scenario], dtype=float)
        out[key] = {"mean": float(np.mean(vals)), "std": float(np.std(vals)), "min": float(np.min(vals)), "max": float(np.max(vals))}
    return out


def evaluate_config(build: Callable[[str], ScenarioConfig], names: List[str], method: str, config: Dict[str, Any],
                    options: Dict[str, Any], **planner_extra) -> Dict[str, Any]:
    rows: List[Dict[str, Any]] = []
    for name in names:
        try:
            rows.append(run_single(build(name), name, method, config, options, **planner_extra))
        except Exception as exc:  # the reference swallows per-scenario failures into a sentinel row
            rows.append({"scenario": name, "runtime_s": 0.0, "score": 1e9, "success": False, "fun": 1e9, "length": 0.0,
                         "min_clearance": -1.0, "nit": 0, "message": f"Exception: {exc}"})
    # sentinel rows lack most numeric keys, exactly as in the reference: the aggregation then raises KeyError
    agg = aggregate_numeric(rows)
    return {"mean_score": float(agg["score"]["mean"]), "std_score": float(agg["score"]["std"]),
            "success_rate": float(np.mean([1.0 if r["success"] else 0.0 for r in rows])), "metrics": agg, "per_scenario": rows}


def hyperopt(build, train: List[str], method: str, n_trials: int, seed: int, **planner_extra) -> Dict[str, Any]:
    rng = np.random.default_rng(seed)
    trials: List[Dict[str, Any]] = []
    best_i, best_score = -1, float("inf")
    for i in range(n_trials):
        cfg, opts = sample_strategy_config(method, rng)
        res = evaluate_config(build, train, method, cfg, opts, **planner_extra)
        trials.append({"trial": i + 1, "method": method, "config": cfg, "options": opts, "mean_score": res["mean_score"],
                       "std_score": res["std_score"], "success_rate": res["success_rate"]})
        if res["mean_score"] < best_score:
            best_score, best_i = res["mean_score"], i
    return {"trials": trials, "best": trials[best_i]}


def method_seed(seed: int, method: str) -> int:
    name = method.upper()
    return seed + (0 if name == "POWELL" else 10_000 if name == "CEM" else 20_000)


def run_sweep(build: Callable[[str], ScenarioConfig], scenario_names: List[str], methods=("Powell", "CEM", "CEM-POWELL"),
              trials: int = 12, seed: int = 7, log: Optional[Callable[[str], None]] = print, **planner_extra) -> Dict[str, Any]:
    """The full loop of the reference's ``main`` (ref :342-397): hyperopt on the train split, then
    benchmark the best configuration of each method on every scenario."""
    names = list(scenario_names)
    if not names:
        raise ValueError("No scenarios found for benchmark.")
    train = names[:-1] if len(names) > 1 else names
    for m in methods:
        if m.upper() not in {"POWELL", "CEM", "CEM-POWELL", "HYBRID"}:
            raise ValueError(f"Unsupported method '{m}'. Use Powell, CEM, and/or CEM-POWELL.")
    hyper, bench = {}, {}
    for method in methods:
        if log:
            log(f"[hyperopt] method={method} trials={trials} train_scenarios={len(train)}")
        h = hyperopt(build, train, method, trials, method_seed(seed, method), **planner_extra)
        hyper[method] = h
        best = h["best"]
        res = evaluate_config(build, names, str(best["method"]), dict(best["config"]), dict(best["options"]), **planner_extra)
        bench[method] = {"method": method, "config": dict(best["config"]), "options": dict(best["options"]),
                         "aggregate": {k: res[k] for k in ("mean_score", "std_score", "success_rate", "metrics")},
                         "per_scenario": res["per_scenario"]}
        if log:
            log(f"[best] method={method} mean_score={best['mean_score']:.4f} success_rate={best['success_rate']:.2f}")
            log(f"[benchmark] method={method} mean_score={res['mean_score']:.4f} std={res['std_score']:.4f} "
                f"success_rate={res['success_rate']:.2f}")
    return {"train_scenarios": train, "test_scenarios": names, "methods": list(methods), "trials_per_method": int(trials),
            "seed": int(seed), "hyperopt": hyper, "benchmark": bench}


def write_results(payload: Dict[str, Any], path) -> None:
    out = Path(path)
    out.parent.mkdir(parents=True, exist_ok=True)
    out.write_text(json.dumps(payload, indent=2), encoding="utf-8")
