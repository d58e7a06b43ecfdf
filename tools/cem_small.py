"""Small device-resident CEM solves (one scenario x 32 restarts): wall time per solve, with and without the captured graph
(MP_CEM_GRAPH=0)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from motion_planning_amd import _native
from motion_planning_amd.synthetic import wall_scenario_problems

probs = wall_scenario_problems(_native.build_basis, 4, seed=5)
for n_sc in (1, 4):
    ps = probs[:n_sc]
    D = ps[0].dim
    ctx = _native.Context(0)
    ctx.set_problem(list(ps))
    rng = np.random.default_rng(0)
    x0 = np.vstack([p.x0[None, :] + p.sigma0[None, :] * rng.normal(size=(32, D)) for p in ps])
    s0 = np.vstack([np.repeat(p.sigma0[None, :], 32, axis=0) for p in ps])
    tof = np.repeat(np.arange(n_sc, dtype=np.int32), 32)
    res = ctx.cem_solve(x0, s0, task_of=tof, population_size=64, max_iter=90, seed=1)
    t0 = time.perf_counter()
    for _ in range(10):
        res = ctx.cem_solve(x0, s0, task_of=tof, population_size=64, max_iter=90, seed=1)
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{n_sc} scenario(s) x 32 restarts: {ms:.2f} ms per solve, nit mean {res['nit'].mean():.1f}, best {res['fun'].min():.6f}, checksum {res['fun'].sum():.9f}")
    ctx.close()
