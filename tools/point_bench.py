"""Point-model shapes (optimize_bspline_with_vias: 3-term objective, point SDF): kernel time per launch."""
import sys, dataclasses
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from motion_planning_amd import _native
from motion_planning_amd.batch import DeviceBatch
from motion_planning_amd.synthetic import synthetic_problem
from tests import helpers as H

case = next(c for c in H.load_cases() if c["name"] == "vias_point")
prob = H.problem_for_case(case)
big = synthetic_problem(_native.build_basis, 50, 64, 0)
n = 65536
rng = np.random.default_rng(0)
for label, p in (("vias_point (3 boxes)", prob), ("vias_point + 50 boxes", dataclasses.replace(prob, boxes=big.boxes))):
    X = torch.from_numpy(p.x0 + p.sigma0 * rng.normal(size=(n, p.dim))).cuda()
    for grad in (False, True):
        db = DeviceBatch(p, n, want_grad=grad)
        for _ in range(2): db.evaluate(X)
        torch.cuda.synchronize()
        db.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        db.ctx.timer_start()
        for _ in range(10): db.evaluate(X)
        ms = db.ctx.timer_stop() / 10
        print(f"{label:26s} S={p.n_samples} grad={grad}: {ms:.3f} ms  {n / ms * 1e3:.3e} evals/s", flush=True)
        db.close()
