#!/usr/bin/env python3
"""BASELINE config 1 on the GPU path: one scenario, DEMO_CONFIG (counterpart of example/demo.py
without the plotting).  Prints the solve time and the cost breakdown the demo prints."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from motion_planning_amd.configs import DEMO_CONFIG  # noqa: E402
from motion_planning_amd.scenarios import WorldModel, scenario_from_resolved  # noqa: E402
from motion_planning_amd.spline_opt import optimize_bspline_path  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenario", default="front")
    ap.add_argument("--scenarios-file", default=str(ROOT / "tests" / "golden" / "scenarios.json"),
                    help="resolved-scenario JSON fixture, or a YAML file in the reference's format")
    ap.add_argument("--method", default=None)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--cprofile", type=int, default=0, help="print the top N functions (cumulative time) of a third, profiled solve")
    args = ap.parse_args()
    path = Path(args.scenarios_file)
    payload = json.loads(path.read_text()) if path.suffix == ".json" else None
    if payload is not None and "meta" in payload:
        if args.scenario not in payload["scenarios"]:
            raise ValueError(f"Unknown scenario '{args.scenario}'. Available: {', '.join(sorted(payload['scenarios']))}")
        sc = scenario_from_resolved(payload["scenarios"][args.scenario])
    else:
        sc = WorldModel(path).build_scenario(args.scenario)
    cfg = dict(DEMO_CONFIG)
    if args.method:
        cfg["method"] = args.method
        cfg.pop("options", None)
    from motion_planning_amd import spline_opt as so

    held = {}

    def factory(problem):
        held["ev"] = so.NativeEvaluator(problem, dtype=np.float64 if args.dtype == "f64" else np.float32)
        return held["ev"]

    def solve():
        t0 = time.perf_counter()
        out = optimize_bspline_path(
            scene=sc.scene, start=sc.start, goal=sc.goal, moving_block_size=sc.moving_block_size,
            start_yaw_deg=sc.start_yaw_deg, goal_yaw_deg=sc.goal_yaw_deg,
            goal_approach_normals=np.asarray(sc.goal_normals, dtype=float), evaluator_factory=factory, **cfg)
        return out, time.perf_counter() - t0

    (S, vias, info), t_first = solve()       # includes HIP initialisation and the first launch of every kernel
    (S, vias, info), t_warm = solve()
    ev = held["ev"]
    if args.cprofile:
        import cProfile
        import pstats

        pr = cProfile.Profile()
        pr.enable()
        solve()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(args.cprofile)
    print(f"Optimization took {t_warm:.4f} seconds ({ev.n_calls} evaluator calls, {ev.n_evals} objective evaluations; "
          f"first solve of the process, GPU initialisation included: {t_first:.2f} s)")
    print(json.dumps({"config": "BASELINE config 1", "scenario": args.scenario, "method": cfg.get("method", "Powell"),
                      "wall_s": t_warm, "wall_first_s": t_first, "nit": int(info["nit"]), "evaluator_calls": ev.n_calls,
                      "objective_evals": ev.n_evals, "fun": float(info["fun"]), "min_clearance": float(info["min_clearance"])}))
    print(f"success={info['success']} nit={info['nit']} fun={info['fun']:.6f} min_clearance={info['min_clearance']:.6f} "
          f"mean_clearance={info['mean_clearance']:.6f}")
    for k in ("length", "curvature_cost", "yaw_smoothness_cost", "safety_cost", "preferred_safety_cost",
              "approach_rebound_cost", "goal_clearance_cost", "goal_clearance_target_cost", "approach_clearance_cost",
              "approach_collision_cost", "via_deviation_cost", "yaw_deviation_cost", "yaw_monotonic_cost",
              "yaw_schedule_cost", "goal_approach_normal_cost", "turn_angle_mean_deg"):
        print(f"  {k:28s} {info[k]:.6f}")
    print("  vias:", np.array2string(vias, precision=4))


if __name__ == "__main__":
    main()
