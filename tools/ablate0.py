"""Fixed machinery of one evaluation: config-3 shape with NO boxes, term groups switched off one by one."""
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
full = p3.term_mask
variants = {"all terms": full, "no curvature": full & ~2, "no closed-form (via/yaw dev, normal)": full & ~((1 << 10) | (1 << 11) | (1 << 14)),
            "no yaw terms": full & ~((1 << 2) | (1 << 12) | (1 << 13)), "no clearance terms": full & ~0x3f8, "no length": full & ~1, "nothing": 0}
for nb in (0, 50):
    for name, m in variants.items():
        p = dataclasses.replace(p3, boxes=p3.boxes[:nb].copy(), term_mask=int(m))
        row = []
        for grad in (False, True):
            db = DeviceBatch(p, n, want_grad=grad)
            for _ in range(2): db.evaluate(X)
            torch.cuda.synchronize()
            db.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            db.ctx.timer_start()
            for _ in range(10): db.evaluate(X)
            row.append(db.ctx.timer_stop() / 10)
            db.close()
        print(f"boxes={nb:2d} {name:40s} cost-only {row[0]:.3f} ms   cost+grad {row[1]:.3f} ms", flush=True)
