#!/usr/bin/env python3
"""Where the cycles of eval_kernel go, per phase (investigation aid; needs the -DMP_PHASES build -> build_ab/libmp_phases.so, cf.
tools/clearance_stats.py).  Cycle deltas are wall cycles of a wave (other waves interleave), so read them as shares."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from motion_planning_amd import _native  # noqa: E402
from motion_planning_amd.synthetic import synthetic_candidates, synthetic_problem  # noqa: E402

NAMES = ["waypoints+spline", "pass 1 (length, yaw, d1)", "curvature", "clearance profile", "clearance terms", "closed-form terms",
         "reductions+outputs", "grad: clearance chain", "grad: curvature^T", "grad: gather", "grad: M^T g"]
B, S, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (50, 64, 65536)))
lib = _native.load()
prob = synthetic_problem(_native.build_basis, B, S, seed=0)
X = synthetic_candidates(prob, N, seed=2)
ctx = _native.Context(0)
ctx.set_problem(prob)
ph = (C.c_ulonglong * 16)()
for grad in (True, False):
    ctx.eval(X, want_grad=grad, want_terms=False)
    lib.mp_debug_phases(ph)
    ctx.eval(X, want_grad=grad, want_terms=False)
    lib.mp_debug_phases(ph)
    tot = sum(ph[i] for i in range(11))
    print(f"B={B} S={S} N={N} grad={grad}: {tot / N:.0f} wave-cycles per candidate")
    for i, nm in enumerate(NAMES):
        if ph[i]:
            print(f"   {nm:28s} {ph[i] / N:9.0f}  {100.0 * ph[i] / tot:5.1f} %")
ctx.close()
