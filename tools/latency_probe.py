#!/usr/bin/env python3
"""Latency of small host-buffer evaluations (what SciPy Powell / a host CEM generation see): n = 1, 8, 64, 512."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from motion_planning_amd import _native  # noqa: E402
from tests import helpers as H  # noqa: E402

case = next(c for c in H.load_cases() if c["name"] == "base_front")
prob = H.problem_for_case(case)
ctx = _native.Context(0)
ctx.set_problem(prob)
rng = np.random.default_rng(0)
for n in (1, 8, 64, 512):
    X = prob.x0 + 0.1 * rng.normal(size=(n, prob.dim))
    for kw in (dict(want_terms=False), dict(want_terms=True), dict(want_terms=True, want_grad=True)):
        for _ in range(20):
            ctx.eval(X, **kw)
        t0 = time.perf_counter()
        reps = 300
        for _ in range(reps):
            ctx.eval(X, **kw)
        us = (time.perf_counter() - t0) / reps * 1e6
        print(f"n={n:4d} {kw}: {us:7.1f} us per call ({us / n:7.2f} us per candidate)", flush=True)
ctx.close()
