#!/usr/bin/env python3
"""Work counters of the culled clearance phase (investigation aid).  Needs the -DMP_STATS build:
  hipcc ... -DMP_STATS motion_planning_amd/csrc/mp_api.hip -o build_ab/libmp_stats.so
  MP_HIP_LIB=build_ab/libmp_stats.so python tools/clearance_stats.py [n_boxes n_samples n_cand]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from motion_planning_amd import _native  # noqa: E402
from motion_planning_amd.synthetic import synthetic_candidates, synthetic_problem  # noqa: E402

B, S, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (2000, 256, 8192)))
lib = _native.load()
prob = synthetic_problem(_native.build_basis, B, S, seed=1 if B >= 1000 else 0, size_range=(0.05, 0.3) if B >= 1000 else (0.1, 0.6))
X = synthetic_candidates(prob, N, seed=1000)
ctx = _native.Context(0)
ctx.set_problem(prob)
st = (C.c_ulonglong * 8)()
lib.mp_debug_stats(st)
ctx.eval(X, want_grad=True)
lib.mp_debug_stats(st)
chunks, groups, boxes, pairs, drains = (st[i] for i in range(5))
print(f"B={B} S={S} N={N}: chunks={chunks} per chunk: groups tested {st[7] / chunks:.1f}, groups visited {groups / chunks:.1f}, boxes bound-tested {boxes / chunks:.1f}, "
      f"pairs evaluated {pairs / chunks:.1f} in {drains / chunks:.2f} drains ({pairs / max(drains, 1):.1f} per drain); "
      f"of the pairs {100.0 * st[5] / max(pairs, 1):.1f} % xy-separated (need the rectangle distance), {100.0 * st[6] / max(pairs, 1):.1f} % overlapping")
ctx.close()
