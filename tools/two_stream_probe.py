#!/usr/bin/env python3
"""Config-3 batches back to back on one stream vs alternating over two streams (two contexts, two output slots):
how much of a launch's ramp-up and tail the next independent batch can hide."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from motion_planning_amd import _native
from motion_planning_amd.batch import DeviceBatch
from motion_planning_amd.synthetic import synthetic_candidates, synthetic_problem

N, STEPS = 65536, 200
prob = synthetic_problem(_native.build_basis, 50, 64, seed=0)
X = torch.from_numpy(synthetic_candidates(prob, N, seed=5)).cuda()
for grad in (True, False):
    for nstreams in (1, 2, 3):
        dbs = [DeviceBatch(prob, N, want_grad=grad) for _ in range(nstreams)]
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(STEPS):
                with torch.cuda.stream(streams[i % nstreams]):
                    dbs[i % nstreams].evaluate(X)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / STEPS
        print(f"grad={grad!s:5} streams={nstreams}: {dt * 1e3:.3f} ms per batch  {N / dt:.3e} evals/s", flush=True)
        for d in dbs:
            d.close()
