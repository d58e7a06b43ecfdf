"""Quick parity + throughput check used while tuning kernels (GPU box)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from motion_planning_amd import _native
from motion_planning_amd.batch import DeviceBatch
from motion_planning_amd.synthetic import synthetic_problem, synthetic_candidates
from oracle import oracle as orc

def check(prob, n, dtype):
    X = synthetic_candidates(prob, n, 1, dtype)
    ctx = _native.Context(0, dtype); ctx.set_problem(prob)
    o = ctx.eval(X, want_grad=True, want_dist=True); r = orc.evaluate(prob, X, dtype=dtype)
    ok = np.array_equal(o["dist"], r["dist"]) and np.array_equal(o["flags"], r["flags"]) and np.array_equal(o["min_clear"], r["min_clear"])
    ce = np.max(np.abs(o["cost"].astype(float) - r["cost"].astype(float)) / np.abs(r["cost"].astype(float)))
    ge = np.max(np.abs(o["grad"].astype(float) - r["grad"].astype(float)) / (np.abs(r["grad"].astype(float)).max(axis=1, keepdims=True)))
    o2 = ctx.eval(X, want_grad=True, want_terms=False)      # the launch the benchmarks time
    ok = ok and np.array_equal(o2["grad"], o["grad"]) and np.array_equal(o2["flags"], o["flags"])
    print(f"  parity {np.dtype(dtype).name} S={prob.n_samples} B={prob.boxes.shape[0]}: exact={ok} cost_err={ce:.2e} grad_err={ge:.2e}")
    assert ok and ce < 1e-4 and ge < 2e-3
    ctx.close()

def bench(prob, n, dtype, grad, steps=10):
    X = torch.from_numpy(synthetic_candidates(prob, n, 2, dtype)).cuda()
    db = DeviceBatch(prob, n, dtype=dtype, want_grad=grad)
    for _ in range(2): db.evaluate(X)
    torch.cuda.synchronize()
    db.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    db.ctx.timer_start()
    for _ in range(steps): db.evaluate(X)
    ms = db.ctx.timer_stop() / steps
    db.close()
    return ms

if __name__ == "__main__":
    p3 = synthetic_problem(_native.build_basis, 50, 64, 0)
    check(p3, 2048, np.float64); check(p3, 2048, np.float32)
    p101 = synthetic_problem(_native.build_basis, 50, 101, 0)
    check(p101, 512, np.float64)
    n = 65536
    for dt in (np.float64, np.float32):
        for g in (True, False):
            ms = bench(p3, n, dt, g)
            print(f"config3 {np.dtype(dt).name} grad={g}: {ms:.3f} ms -> {n/ms*1e3:.3e} evals/s")
    if len(sys.argv) > 1:
        p4 = synthetic_problem(_native.build_basis, 2000, 256, 1, (0.05, 0.3))
        check(p4, 128, np.float64)
        ms = bench(p4, 8192, np.float64, True, steps=3)
        print(f"config4-shaped fp64 grad: {ms:.3f} ms / 8192 -> {8192/ms*1e3:.3e} evals/s")

    # host-buffer path of mp_eval (PCIe + staging inclusive), for DESIGN.md §7
    import time
    ctx = _native.Context(0, np.float64); ctx.set_problem(p3)
    Xh = synthetic_candidates(p3, n, 2)
    ctx.eval(Xh, want_terms=False, want_grad=True)
    t0 = time.perf_counter()
    for _ in range(5): ctx.eval(Xh, want_terms=False, want_grad=True)
    dt = (time.perf_counter() - t0) / 5
    print(f"config3 fp64 grad, HOST buffers in/out: {dt*1e3:.3f} ms -> {n/dt:.3e} evals/s")
