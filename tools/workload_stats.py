#!/usr/bin/env python3
"""Exact pair evaluations per candidate for every case of bench.py's workload sweep (-DMP_STATS build of the library:
`tools/build_variant.sh stats -DMP_STATS`, then `MP_HIP_LIB=ab/libmp_stats.so python tools/workload_stats.py`).
Writes profiles/r03_workload_stats.json, which bench.py merges into extras.workload_sweep."""
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from motion_planning_amd import _native  # noqa: E402
from motion_planning_amd.synthetic import synthetic_candidates, synthetic_problem  # noqa: E402

lib = _native.load()
assert hasattr(lib, "mp_debug_stats"), "needs the -DMP_STATS build (MP_HIP_LIB=ab/libmp_stats.so)"
problem = synthetic_problem(_native.build_basis, bench.N_BOXES, bench.N_SAMPLES, seed=0)
Xh = synthetic_candidates(problem, bench.N_CAND, seed=100)
st = (C.c_ulonglong * 8)()
out = {}
cases = [("headline", problem, Xh)] + list(bench.workload_cases(_native, problem, Xh, np.float64))
for name, prob, X in cases:
    ctx = _native.Context(0)
    ctx.set_problem(prob)
    lib.mp_debug_stats(st)
    n = 8192      # the counters are contended atomics: a strided sample of the batch is enough
    res = ctx.eval(np.ascontiguousarray(X[:: len(X) // n][:n]), want_grad=True, want_terms=False)
    lib.mp_debug_stats(st)
    chunks, groups, boxes, pairs, drains = (int(st[i]) for i in range(5))
    out[name] = {"exact_pairs_per_candidate": pairs / n, "drains_per_candidate": drains / n, "boxes_bound_tested_per_candidate": boxes / n,
                 "xy_separated_fraction": st[5] / max(pairs, 1), "overlapping_fraction": st[6] / max(pairs, 1),
                 "colliding_fraction": float((res["flags"] & 1).mean()), "sample": n}
    print(name, out[name], flush=True)
    ctx.close()
(ROOT / "profiles" / "r03_workload_stats.json").write_text(json.dumps(out, indent=1))
