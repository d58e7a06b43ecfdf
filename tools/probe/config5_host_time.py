#!/usr/bin/env python3
"""Where the host time of the config-5 job goes (GPU box): restart offsets, Context.set_problem (Python structs + C upload), the
device-resident solve, the result handling.  usage: python tools/probe/config5_host_time.py [scenarios]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch  # noqa: F401,E402
from motion_planning_amd import _abi as abi, _native  # noqa: E402
from motion_planning_amd._native import as_boxes10  # noqa: E402
from motion_planning_amd.synthetic import wall_scenario_problems  # noqa: E402

n_s = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
probs = wall_scenario_problems(_native.build_basis, n_s, seed=2)
restarts, D = 32, probs[0].dim
torch.zeros(1).cuda()
t = time.perf_counter()
x0 = np.empty((n_s, restarts, D)); s0 = np.empty((n_s, restarts, D))
for i, p in enumerate(probs):
    eps = np.random.default_rng([2, i]).normal(size=(restarts, D)); eps[0] = 0.0
    x0[i] = p.x0[None, :] + eps * p.sigma0[None, :]; s0[i] = p.sigma0[None, :]
print(f"restart offsets (Python loop)      {time.perf_counter() - t:.3f} s")
t = time.perf_counter(); ctx = _native.Context(0); print(f"Context()                          {time.perf_counter() - t:.3f} s")
t = time.perf_counter(); b = np.vstack([as_boxes10(p.boxes) for p in probs]); print(f"  vstack of the boxes              {time.perf_counter() - t:.3f} s")
t = time.perf_counter(); ctx.set_scene(b); print(f"  set_scene                        {time.perf_counter() - t:.3f} s")
t = time.perf_counter()
cfg = probs[0].config_struct(); tasks = (abi.MpTask * n_s)(); off = 0
for i, p in enumerate(probs):
    tasks[i] = p.task_struct(box_begin=off, box_count=p.boxes.shape[0]); off += p.boxes.shape[0]
print(f"  task structs (Python loop)       {time.perf_counter() - t:.3f} s")
t = time.perf_counter(); ctx.set_problem(probs, upload_scene=False); print(f"  set_problem (structs again + C)  {time.perf_counter() - t:.3f} s")
task_of = np.repeat(np.arange(n_s, dtype=np.int32), restarts)
t = time.perf_counter()
res = ctx.cem_solve(x0.reshape(-1, D), s0.reshape(-1, D), task_of=task_of, population_size=64, max_iter=90, seed=2)
print(f"cem_solve                          {time.perf_counter() - t:.3f} s")
ctx.close()
