"""Ablation at the planner's own shape (S=101, few boxes, cost only = a CEM population): boxes 0/1/4/13, term groups."""
import sys, dataclasses
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from motion_planning_amd import _native
from motion_planning_amd.batch import DeviceBatch
from motion_planning_amd.synthetic import synthetic_problem, synthetic_candidates

p3 = synthetic_problem(_native.build_basis, 13, 101, 0)
n = 131072
X = torch.from_numpy(synthetic_candidates(p3, n, 2)).cuda()
def run(p, grad, label):
    db = DeviceBatch(p, n, want_grad=grad)
    for _ in range(2): db.evaluate(X)
    torch.cuda.synchronize()
    db.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    db.ctx.timer_start()
    for _ in range(10): db.evaluate(X)
    ms = db.ctx.timer_stop() / 10
    print(f"{label:44s} grad={grad}: {ms:.3f} ms  {n / ms * 1e3:.3e} evals/s", flush=True)
    db.close()
for nb in (0, 1, 4, 13):
    run(dataclasses.replace(p3, boxes=p3.boxes[:nb].copy()), False, f"boxes={nb}")
full = p3.term_mask
for name, m in {"no curvature": full & ~2, "no via/yaw dev, goal normal": full & ~((1 << 10) | (1 << 11) | (1 << 14)),
                "no yaw smooth/mono/sched": full & ~((1 << 2) | (1 << 12) | (1 << 13)), "clearance terms only": full & 0x3f8,
                "nothing": 0}.items():
    run(dataclasses.replace(p3, term_mask=int(m)), False, name)
run(p3, True, "all terms, 13 boxes")
