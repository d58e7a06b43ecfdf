#!/usr/bin/env python3
"""BASELINE config 4: cluttered synthetic scene — 2,000 boxes, 256 samples/trajectory, 1,048,576 candidates
sharded by index over the GPUs of the job (131,072 per GPU on 8 GPUs), fp64 cost + gradient.

  python tools/run_config4.py                     # one GPU's shard of the 8-GPU job
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/run_config4.py --total 1048576
"""
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
    ap.add_argument("--total", type=int, default=1_048_576)
    ap.add_argument("--shards", type=int, default=8, help="shard count the total is divided by when run on fewer GPUs")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    from motion_planning_amd import _native
    from motion_planning_amd.batch import DeviceBatch, allgather_results, pack_results, shard_bounds
    from motion_planning_amd.synthetic import synthetic_candidates, synthetic_problem

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    shards = world if world > 1 else args.shards
    lo, hi = shard_bounds(args.total, rank if world > 1 else 0, shards)
    n = hi - lo
    dt = np.float64 if args.dtype == "f64" else np.float32
    prob = synthetic_problem(_native.build_basis, 2000, 256, seed=1, size_range=(0.05, 0.3))
    X = torch.from_numpy(synthetic_candidates(prob, n, seed=1000 + rank, dtype=dt)).cuda()
    db = DeviceBatch(prob, n, device=local if world > 1 else 0, dtype=dt, want_grad=True)
    db.evaluate(X)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = db.evaluate(X)
    if world > 1:
        full = allgather_results(pack_results(out["cost"], out["min_clear"], out["flags"]), args.total, rank, world)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = (time.perf_counter() - t0) / args.steps
    if rank == 0:
        fl = out["flags"].cpu().numpy()
        flops = (256 * 2000 * 200 + 256 * 160 + 16 * 256 * prob.n_ctrl) * n * shards if world == 1 else None
        print(json.dumps({"metric": "spline cost+grad evals/sec", "value": n * (world if world > 1 else 1) / el, "unit": "evals/s",
                          "n_gpus": world, "candidates_this_gpu": n, "total_candidates": args.total, "s_per_pass": el,
                          "dtype": args.dtype, "colliding_fraction": float((fl & 1).mean()),
                          "config": {"workload": "BASELINE config 4: 2,000 boxes, 256 samples, 1M candidates sharded by index",
                                     "algorithmic_flops_per_eval": 256 * 2000 * 200 + 256 * 160 + 16 * 256 * prob.n_ctrl}}))
    db.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
