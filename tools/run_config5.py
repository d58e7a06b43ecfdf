#!/usr/bin/env python3
"""BASELINE config 5: scenario-parallel planning — N generated scenarios x R restarts, each a full
CEM solve to the reference's stopping rule, all instances in lock-step on the GPU(s).

  python tools/run_config5.py --scenarios 10000 --restarts 32
  python tools/run_config5.py --gpus G ...                    # starts the G ranks itself; or under torchrun:
  python -m torch.distributed.run --nproc-per-node G --master-addr 127.0.0.1 tools/run_config5.py ...
Scenarios are sharded by index in contiguous blocks (batch.shard_bounds); the only exchange is one all-gather of
per-scenario (best cost, solved flag, generations).  Restart offsets and device noise streams are keyed by the GLOBAL
scenario index, so the gathered table does not depend on the number of ranks (`--dump` writes it for a checker)."""
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
    ap.add_argument("--gpus", type=int, default=1, help="> 1 without torchrun: this script starts that many ranks itself")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--one-device", action="store_true", help="every rank uses cuda:0 (rehearsal on a one-GPU box)")
    ap.add_argument("--dump", default=None, help="rank 0 writes the gathered per-scenario table to this .npy")
    args = ap.parse_args()
    # --gpus N without a launcher: start the N ranks as a fresh child process before anything here touches the GPU
    from motion_planning_amd import launch

    if args.gpus > 1 and not launch.under_launcher():
        sys.exit(launch.self_launch(str(Path(__file__).resolve()), sys.argv[1:], args.gpus, one_device=args.one_device))
    if launch.under_launcher() and args.gpus > 1 and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks")
    import torch
    import torch.distributed as dist

    from motion_planning_amd import _native
    from motion_planning_amd.batch import shard_bounds, solve_scenarios_cem
    from motion_planning_amd.synthetic import wall_scenario_problems

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
        dist.barrier()      # RCCL's first collective takes tens of ms: not inside the timed solve
    t_build = time.perf_counter()
    probs = wall_scenario_problems(_native.build_basis, args.scenarios, seed=args.seed)
    lo, hi = shard_bounds(args.scenarios, rank, world)
    mine = probs[lo:hi]
    t_build = time.perf_counter() - t_build
    dt = np.float64 if args.dtype == "f64" else np.float32
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    res = solve_scenarios_cem(mine, restarts=args.restarts, device=local if world > 1 else 0, dtype=dt,
                              population_size=args.population, max_iter=args.max_iter, seed=args.seed, restart_seed=args.seed,
                              scenario_base=lo)
    # the path's only exchange: ONE all-gather of per-scenario (best cost, solved flag, generations), shards padded to equal length
    width = max(b - a for a, b in (shard_bounds(args.scenarios, r, world) for r in range(world)))
    rows = torch.zeros(width, 3, dtype=torch.float64, device="cuda")
    rows[:, 1] = -1.0                                   # padding rows: no scenario
    k = len(mine)
    rows[:k, 0] = torch.as_tensor(np.asarray(res["fun"], dtype=np.float64)).cuda()
    rows[:k, 1] = torch.as_tensor(np.asarray(res["solved"], dtype=np.float64)).cuda()
    rows[:k, 2] = torch.as_tensor(np.asarray(res["nit"], dtype=np.float64).sum(axis=1)).cuda()
    if world > 1:
        src = rows if args.backend == "nccl" else rows.cpu()     # gloo moves host memory
        gathered = torch.empty(world * width, 3, dtype=torch.float64, device=src.device)
        dist.all_gather_into_tensor(gathered, src)
        torch.cuda.synchronize()
        dist.barrier()
        parts = []
        for r in range(world):
            a, b = shard_bounds(args.scenarios, r, world)
            parts.append(gathered[r * width: r * width + (b - a)])
        full = torch.cat(parts, dim=0).cpu()
    else:
        full = rows[: args.scenarios].cpu()
    assert bool((full[:, 1] >= 0).all()), "a scenario is missing from the gathered table"
    tot = torch.tensor([float(full[:, 1].sum()), float(full.shape[0]), float(full[:, 2].sum()) * args.population])
    elapsed = time.perf_counter() - t0
    if rank == 0 and args.dump:
        np.save(args.dump, full.numpy())
    if rank == 0:
        n_solved, n_all, evals = (float(v) for v in tot.cpu())
        print(json.dumps({
            "metric": "scenarios solved/sec", "value": n_solved / elapsed, "unit": "scenarios/s", "n_gpus": world,
            "scenarios": int(n_all), "restarts": args.restarts, "solved": int(n_solved), "solved_fraction": n_solved / n_all,
            "wall_s": elapsed, "objective_evals": evals, "evals_per_s": evals / elapsed, "dtype": args.dtype,
            "scenario_build_s": t_build, "best_cost_sum_of_solved": float(full[full[:, 1] > 0, 0].sum()),
            "config": {"workload": "BASELINE config 5: generated wall-building scenarios x restarts, CEM to the reference "
                                   "stopping rule, S=101, BASE_CONFIG", "population": args.population, "max_iter": args.max_iter},
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
