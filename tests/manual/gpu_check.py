"""Ad-hoc GPU check: HIP path vs CPU oracle on demo-shaped and config-3-shaped problems."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from motion_planning_amd import _native
from motion_planning_amd.blocks import Block
from motion_planning_amd.configs import BASE_CONFIG
from motion_planning_amd.problem import build_path_problem, build_vias_problem
from motion_planning_amd.synthetic import BoxList, synthetic_problem, synthetic_candidates
from oracle import oracle as orc


def report(name, got, ref, exact=False):
    got = np.asarray(got); ref = np.asarray(ref)
    if exact:
        ok = np.array_equal(got, ref)
        print(f"  {name:10s} exact={ok}  mismatches={int(np.sum(got != ref))}/{got.size}")
        return ok
    with np.errstate(invalid="ignore", divide="ignore"):
        err = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-12)
    err = np.where(np.isfinite(err), err, 0.0)
    print(f"  {name:10s} max rel err {err.max():.3e}")
    return err.max()


def run(problem, n, dtype, label):
    print(f"== {label}  dtype={np.dtype(dtype).name} n={n} S={problem.n_samples} B={problem.boxes.shape[0]}")
    ctx = _native.Context(0, dtype)
    ctx.set_problem(problem)
    X = synthetic_candidates(problem, n, seed=1, dtype=dtype)
    t0 = time.time(); out = ctx.eval(X, want_grad=True, want_dist=True); t1 = time.time()
    ref = orc.evaluate(problem, X, dtype=dtype); t2 = time.time()
    print(f"  gpu {t1-t0:.3f}s  oracle {t2-t1:.3f}s")
    report("dist", out["dist"], ref["dist"], exact=True)
    report("min_clear", out["min_clear"], ref["min_clear"], exact=True)
    report("flags", out["flags"], ref["flags"], exact=True)
    report("cost", out["cost"], ref["cost"])
    report("terms", out["terms"], ref["terms"])
    gerr = np.abs(out["grad"] - ref["grad"]) / (np.abs(ref["grad"]) + 1e-6 * np.abs(ref["grad"]).max(axis=1, keepdims=True) + 1e-300)
    print(f"  grad       max rel err {gerr.max():.3e}")
    ctx.close()


if __name__ == "__main__":
    basis = _native.build_basis
    blocks = [Block((2, 2, .1), (0, 0, .05), object_id='table'), Block((.1, 2, 2), (1, 0, 1), object_id='wall'),
              Block((.6, .9, .6), (0, -.5, .4), object_id='cube_top')]
    demo = build_path_problem(BoxList(blocks), (-0.4, -0.2, 1.5), (0.0, 0.25, 0.4), basis, moving_block_size=(.6, .9, .6),
                              start_yaw_deg=0., goal_yaw_deg=90., goal_approach_normals=np.array([[1., 0, 0]]), **BASE_CONFIG)
    run(demo, 512, np.float64, "demo scenario 'front'")
    run(demo, 512, np.float32, "demo scenario 'front'")
    p3 = synthetic_problem(basis, 50, 64, 0)
    run(p3, 2048, np.float64, "config 3")
    run(p3, 2048, np.float32, "config 3")
    pv = build_vias_problem(BoxList(blocks), (-0.4, -0.2, 1.5), (-0.2, 0.5, 1.2), (0.0, 0.25, 0.8), basis,
                            n_additional_vias=2, n_samples_curve=121, collision_check_subsample=7)
    run(pv, 256, np.float64, "3-term point model")
    # throughput, device resident
    import torch
    ctx = _native.Context(0, np.float64); ctx.set_problem(p3)
    n = 65536
    X = torch.from_numpy(synthetic_candidates(p3, n, 2)).cuda()
    cost = torch.empty(n, dtype=torch.float64, device='cuda'); grad = torch.empty(n, p3.dim, dtype=torch.float64, device='cuda')
    minc = torch.empty(n, dtype=torch.float64, device='cuda'); flags = torch.empty(n, dtype=torch.uint8, device='cuda')
    for g in (0, grad.data_ptr()):
        for _ in range(2):
            ctx.eval_device(n, X.data_ptr(), cost.data_ptr(), 0, g, minc.data_ptr(), flags.data_ptr())
        ctx.synchronize()
        ctx.timer_start()
        for _ in range(5):
            ctx.eval_device(n, X.data_ptr(), cost.data_ptr(), 0, g, minc.data_ptr(), flags.data_ptr())
        ms = ctx.timer_stop() / 5
        print(f"config3 fp64 grad={'yes' if g else 'no'}: {ms:.3f} ms / 65536 evals -> {n/ms*1e3:.3e} evals/s")
