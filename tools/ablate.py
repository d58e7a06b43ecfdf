"""Where does the time go?  Same problem with 0, 1, 50 boxes, with / without gradient."""
import sys, dataclasses
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from motion_planning_amd import _native
from motion_planning_amd.batch import DeviceBatch
from motion_planning_amd.synthetic import synthetic_problem, synthetic_candidates

p3 = synthetic_problem(_native.build_basis, 50, 64, 0)
n = 65536
X = torch.from_numpy(synthetic_candidates(p3, n, 2)).cuda()
for nb in (0, 1, 8, 50):
    p = dataclasses.replace(p3, boxes=p3.boxes[:nb].copy())
    for grad in (False, True):
        db = DeviceBatch(p, n, want_grad=grad)
        for _ in range(2): db.evaluate(X)
        torch.cuda.synchronize()
        db.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        db.ctx.timer_start()
        for _ in range(10): db.evaluate(X)
        ms = db.ctx.timer_stop() / 10
        print(f"boxes={nb:3d} grad={grad}: {ms:.3f} ms")
        db.close()

# term ablation at 50 boxes: which cost terms cost time (bit k = term k of _abi.TERM_NAMES)
full = p3.term_mask
variants = {"all terms": full, "no curvature": full & ~(1 << 1), "no via/yaw dev, goal normal": full & ~((1 << 10) | (1 << 11) | (1 << 14)),
            "no yaw smooth/mono/sched": full & ~((1 << 2) | (1 << 12) | (1 << 13)), "no length": full & ~1,
            "clearance terms only": full & 0x3f8, "nothing": 0}
for name, m in variants.items():
    p = dataclasses.replace(p3, term_mask=int(m))
    for grad in (False, True):
        db = DeviceBatch(p, n, want_grad=grad)
        for _ in range(2): db.evaluate(X)
        torch.cuda.synchronize()
        db.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        db.ctx.timer_start()
        for _ in range(10): db.evaluate(X)
        ms = db.ctx.timer_stop() / 10
        print(f"{name:32s} (mask {m:#06x}) grad={grad}: {ms:.3f} ms")
        db.close()
