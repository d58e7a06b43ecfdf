"""Small fixed workload for rocprofv3 counter passes: config 3, fp64, 4 launches with and 4 without gradient
(another shape: MP_PROF_SHAPE="boxes,samples,candidates")."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from motion_planning_amd import _native
from motion_planning_amd.batch import DeviceBatch
from motion_planning_amd.synthetic import synthetic_problem, synthetic_candidates

import os
B, S, n = (int(v) for v in os.environ.get("MP_PROF_SHAPE", "50,64,65536").split(","))
p3 = synthetic_problem(_native.build_basis, B, S, 0)
X = torch.from_numpy(synthetic_candidates(p3, n, 2)).cuda()
for grad in (True, False):
    db = DeviceBatch(p3, n, want_grad=grad)
    for _ in range(4):
        db.evaluate(X)
    torch.cuda.synchronize()
    db.close()
print("done")
