#!/usr/bin/env python3
"""Kernel time of one mp_eval launch for a list of (boxes, samples, candidates) shapes (cost+grad and cost only; fp64, or fp32
with MP_SHAPE_DTYPE=f32)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from motion_planning_amd import _native  # noqa: E402
from motion_planning_amd.batch import DeviceBatch  # noqa: E402
from motion_planning_amd.synthetic import synthetic_candidates, synthetic_problem  # noqa: E402

shapes = [(50, 64, 65536), (50, 101, 32768), (13, 101, 65536), (200, 64, 32768), (2000, 64, 16384), (50, 256, 16384), (2000, 256, 16384)]
if len(sys.argv) > 1:   # e.g. "50,64,65536;13,101,65536"
    shapes = [tuple(int(v) for v in t.split(",")) for t in sys.argv[1].split(";")]
DT = np.float32 if __import__("os").environ.get("MP_SHAPE_DTYPE") == "f32" else np.float64
for B, S, N in shapes:
    prob = synthetic_problem(_native.build_basis, B, S, seed=1 if B >= 1000 else 0, size_range=(0.05, 0.3) if B >= 1000 else (0.1, 0.6))
    X = torch.from_numpy(synthetic_candidates(prob, N, seed=5).astype(DT)).cuda()
    for grad in (True, False):
        db = DeviceBatch(prob, N, want_grad=grad, dtype=DT)
        db.evaluate(X)
        db.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        reps = int(__import__("os").environ.get("MP_SHAPE_REPS", "5"))   # 5: quick A/B; 100+: settled clocks (absolute figures)
        for _ in range(reps if reps > 5 else 0):
            db.evaluate(X)
        db.ctx.timer_start()
        for _ in range(reps):
            db.evaluate(X)
        ms = db.ctx.timer_stop() / reps
        print(f"B={B:5d} S={S:4d} N={N:6d} grad={grad!s:5}: {ms:8.3f} ms  {N / ms * 1e3:.3e} evals/s", flush=True)
        db.close()
