#!/usr/bin/env python3
"""Work counters of the clearance phase over a bounded config-5 CEM solve (investigation aid; -DMP_STATS build):
  MP_HIP_LIB=ab/libmp_stats.so [MP_SEEDS=0] python tools/cem_stats.py [scenarios restarts]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from motion_planning_amd import _native  # noqa: E402
from motion_planning_amd.batch import solve_scenarios_cem  # noqa: E402
from motion_planning_amd.synthetic import wall_scenario_problems  # noqa: E402

n_sc, restarts = (int(v) for v in (sys.argv[1:3] if len(sys.argv) >= 3 else (300, 8)))
lib = _native.load()
probs = wall_scenario_problems(_native.build_basis, n_sc, seed=2)
st = (C.c_ulonglong * 8)()
lib.mp_debug_stats(st)
res = solve_scenarios_cem(probs, restarts=restarts, device=0, dtype=np.float64, population_size=64, max_iter=90, seed=2, restart_seed=2)
lib.mp_debug_stats(st)
chunks, groups, boxes, pairs, drains = (st[i] for i in range(5))
print(f"config-5 CEM, {n_sc} scenarios x {restarts} restarts: chunks={chunks}; per chunk: boxes bound-tested {boxes / chunks:.1f}, "
      f"pairs evaluated {pairs / chunks:.1f} in {drains / chunks:.2f} exact passes ({pairs / max(drains, 1):.1f} per pass); "
      f"solved {int(np.sum(res['solved']))}")
