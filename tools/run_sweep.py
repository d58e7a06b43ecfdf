#!/usr/bin/env python3
"""BASELINE config 2 on the GPU path: the generated-scenario sweep of example/benchmark_planners.py
(3 methods x 12 random-search trials on the train split + benchmark of the best configuration)."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from motion_planning_amd import sweep  # noqa: E402
from motion_planning_amd.scenarios import WorldModel, scenario_from_resolved  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenarios-file", default=str(ROOT / "tests" / "golden" / "scenarios.json"))
    ap.add_argument("--methods", default="Powell,CEM,CEM-POWELL")
    ap.add_argument("--trials", type=int, default=12)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--scenarios", default="")
    ap.add_argument("--output", default=str(ROOT / "gpurun_out" / "benchmark_results.json"))
    args = ap.parse_args()
    path = Path(args.scenarios_file)
    payload = json.loads(path.read_text()) if path.suffix == ".json" else None
    if payload is not None and "meta" in payload:
        table = {k: v for k, v in payload["scenarios"].items() if v.get("file", "").startswith("generated")}
        names = sorted(table)
        build = lambda name: scenario_from_resolved(table[name])  # noqa: E731
    else:
        wm = WorldModel(path)
        names = wm.list_scenarios()
        build = wm.build_scenario
    if args.scenarios.strip():
        names = [s.strip() for s in args.scenarios.split(",") if s.strip()]
    methods = [m.strip() for m in args.methods.split(",") if m.strip()]
    t0 = time.perf_counter()
    out = sweep.run_sweep(build, names, methods, args.trials, args.seed)
    dt = time.perf_counter() - t0
    solves = len(methods) * (args.trials * max(len(names) - 1, 1) + len(names))
    out["scenarios_file"] = str(path)
    out["wall_s"] = dt
    out["solves"] = solves
    sweep.write_results(out, args.output)
    print(f"{solves} solves in {dt:.2f} s -> {solves / dt:.2f} scenarios solved/s; wrote {args.output}")


if __name__ == "__main__":
    main()
