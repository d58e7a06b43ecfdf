#!/usr/bin/env python3
"""BASELINE config 5: scenario-parallel planning — N generated scenarios x R restarts, each a full
CEM solve to the reference's stopping rule, all instances in lock-step on the GPU(s).

  python tools/run_config5.py --scenarios 10000 --restarts 32
  python -m torch.distributed.run --nproc-per-node G --master-addr 127.0.0.1 tools/run_config5.py ...
Scenarios are sharded by index (k mod G); the only exchange is one all-gather of per-scenario
(best cost, solved flag)."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenarios", type=int, default=10000)
    ap.add_argument("--restarts", type=int, default=32)
    ap.add_argument("--population", type=int, default=64)
    ap.add_argument("--max-iter", type=int, default=90)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    from motion_planning_amd import _native
    from motion_planning_amd.batch import solve_scenarios_cem
    from motion_planning_amd.synthetic import wall_scenario_problems

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    t_build = time.perf_counter()
    probs = wall_scenario_problems(_native.build_basis, args.scenarios, seed=args.seed)
    mine = probs[rank::world]
    t_build = time.perf_counter() - t_build
    dt = np.float64 if args.dtype == "f64" else np.float32
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    res = solve_scenarios_cem(mine, restarts=args.restarts, device=local if world > 1 else 0, dtype=dt,
                              population_size=args.population, max_iter=args.max_iter, seed=args.seed + 1000 * rank)
    solved_local = torch.tensor([float(res["solved"].sum()), float(len(mine)), float(res["nit"].sum()) * args.population],
                                dtype=torch.float64, device="cuda")
    if world > 1:
        gathered = [torch.empty_like(solved_local) for _ in range(world)]
        dist.all_gather(gathered, solved_local)
        tot = torch.stack(gathered).sum(dim=0)
        torch.cuda.synchronize()
        dist.barrier()
    else:
        tot = solved_local
    elapsed = time.perf_counter() - t0
    if rank == 0:
        n_solved, n_all, evals = (float(v) for v in tot.cpu())
        print(json.dumps({
            "metric": "scenarios solved/sec", "value": n_solved / elapsed, "unit": "scenarios/s", "n_gpus": world,
            "scenarios": int(n_all), "restarts": args.restarts, "solved": int(n_solved), "solved_fraction": n_solved / n_all,
            "wall_s": elapsed, "objective_evals": evals, "evals_per_s": evals / elapsed, "dtype": args.dtype,
            "scenario_build_s": t_build,
            "config": {"workload": "BASELINE config 5: generated wall-building scenarios x restarts, CEM to the reference "
                                   "stopping rule, S=101, BASE_CONFIG", "population": args.population, "max_iter": args.max_iter},
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
