import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from motion_planning_amd import _native
from motion_planning_amd.batch import DeviceBatch
from motion_planning_amd.synthetic import synthetic_candidates, synthetic_problem
prob = synthetic_problem(_native.build_basis, 50, 64, seed=0)
for N in (16384, 65536, 262144, 1048576):
    X = torch.from_numpy(synthetic_candidates(prob, N, seed=5)).cuda()
    for grad in (True, False):
        db = DeviceBatch(prob, N, want_grad=grad)
        db.evaluate(X)
        db.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        reps = max(5, int(150e6 / N))      # >= 0.1 s of work per measurement, after as much warm-up (clocks settle)
        for _ in range(reps):
            db.evaluate(X)
        db.ctx.timer_start()
        for _ in range(reps):
            db.evaluate(X)
        ms = db.ctx.timer_stop() / reps
        print(f"N={N:8d} grad={grad!s:5}: {ms:8.3f} ms  {N / ms * 1e3:.3e} evals/s", flush=True)
        db.close()
