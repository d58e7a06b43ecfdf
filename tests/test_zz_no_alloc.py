"""mp_eval_noalloc (include/mp_api.h): the device-pointer call that allocates nothing — its split-mode tables live in a
file-scope __device__ array of the library.  Called through the C ABI and checked against mp_eval on the same inputs,
two calls in a row, the second with a smaller input (the array keeps the first call's contents).  Collected last on
purpose (the round's end runs pytest -x)."""
import ctypes as C

import numpy as np
import pytest

from motion_planning_amd import _abi as abi

pytestmark = pytest.mark.gpu

MP_NOALLOC_SCRATCH_BYTES = 32 << 20


def _run_pair(native, prob, dtype, sizes, seeds):
    import torch

    from motion_planning_amd.synthetic import synthetic_candidates

    tdt = torch.float64 if dtype is np.float64 else torch.float32
    ctx = native.Context(0, dtype)
    ctx.set_problem(prob)
    D = prob.dim
    vp = C.c_void_p
    for n, seed in zip(sizes, seeds):
        X = torch.from_numpy(synthetic_candidates(prob, n, seed=seed, dtype=dtype)).cuda()
        ref = {"cost": torch.empty(n, dtype=tdt, device="cuda"), "grad": torch.empty(n, D, dtype=tdt, device="cuda"),
               "minc": torch.empty(n, dtype=tdt, device="cuda"), "flags": torch.empty(n, dtype=torch.uint8, device="cuda")}
        got = {k: torch.full_like(v, 3) for k, v in ref.items()}
        # the existing entry point (its own choice of fused / split kernels, hipMalloc'ed tables when split)
        rc = ctx.lib.mp_eval(ctx._h, n, vp(X.data_ptr()), None, vp(ref["cost"].data_ptr()), None, vp(ref["grad"].data_ptr()),
                             vp(ref["minc"].data_ptr()), vp(ref["flags"].data_ptr()), None, abi.MP_MEM_DEVICE)
        assert rc == 0, ctx.lib.mp_last_error(ctx._h)
        # the entry point under test, straight through the C ABI
        rc = ctx.lib.mp_eval_noalloc(ctx._h, n, vp(X.data_ptr()), None, vp(got["cost"].data_ptr()), vp(got["grad"].data_ptr()),
                                     vp(got["minc"].data_ptr()), vp(got["flags"].data_ptr()))
        assert rc == 0, ctx.lib.mp_last_error(ctx._h)
        ctx.synchronize()
        # bar of the path: flags / clearances bit-exact, cost / gradient 1e-9 (fp64) — here everything is the same
        # arithmetic in the same order, so the comparison is for equality
        for k in ref:
            assert torch.equal(got[k], ref[k]), (k, n)
        assert int((ref["flags"] & 1).sum()) > 0 and torch.isfinite(ref["cost"]).all()
    return ctx


@pytest.mark.parametrize("shape", [(50, 64), (300, 256)], ids=["fused_shape_50x64", "split_shape_300x256"])
def test_noalloc_entry_point_equals_mp_eval(shape):
    from motion_planning_amd import _native as native
    from motion_planning_amd.synthetic import synthetic_problem

    n_boxes, n_samples = shape
    prob = synthetic_problem(native.build_basis, n_boxes, n_samples, seed=0, size_range=(0.05, 0.3) if n_boxes > 100 else (0.1, 0.6))
    for dtype in (np.float64, np.float32):
        # second call smaller than the first: stale rows of the first call stay in the array and must not be read
        ctx = _run_pair(native, prob, dtype, sizes=(4096, 1500), seeds=(5, 6))
        ctx.close()


@pytest.mark.parametrize("shape", [(40, 64, False), (30, 201, True), (25, 255, False), (150, 101, True)],
                         ids=["two_tasks_64", "rotated_wide_odd_201", "wide_odd_255", "rotated_grouped_101"])
def test_noalloc_with_tasks_rotations_and_the_wide_instance(shape):
    """Round-2 advisor finding: the header promises mp_eval's bits for every box-model launch, so cover what the first test
    does not — a device-resident task_of over two tasks with an out-of-range entry (NaN row, no flag), yaw-rotated statics
    (clearance_kernel<T, 1>), and S > 176 with an odd sample count (the wide instance reading the static array), fp64 and
    fp32; two calls in a row, the second smaller."""
    import torch

    from motion_planning_amd import _native as native
    from motion_planning_amd.problem import build_path_problem
    from motion_planning_amd.synthetic import random_scene
    from tests import helpers as H

    n_boxes, S, rotated = shape
    start, goal = np.array([-1.5, -1.5, 1.5]), np.array([1.5, 1.5, 0.4])
    probs = []
    for t in range(2):
        p = build_path_problem(random_scene(n_boxes, 40 + t, (0.1, 0.6), start, goal), start, goal, native.build_basis, n_vias=2,
                               n_samples_curve=S, moving_block_size=(0.9, 0.6, 0.6), start_yaw_deg=0.0, goal_yaw_deg=90.0,
                               combined_4d=True, n_yaw_vias=2, goal_approach_normals=np.array([[0.0, 0.0, 1.0]]))
        probs.append(H.with_rotated_statics(p, seed=t) if rotated else p)
    vp = C.c_void_p
    for dtype in (np.float64, np.float32):
        tdt = torch.float64 if dtype is np.float64 else torch.float32
        ctx = native.Context(0, dtype)
        ctx.set_problem(probs)
        D = probs[0].dim
        for n, seed in ((600, 11), (170, 12)):
            rng = np.random.default_rng(seed)
            tof_h = rng.integers(0, 2, size=n).astype(np.int32)
            tof_h[5], tof_h[n - 3] = 2, -1          # outside the problem: NaN results, no flag, neighbours untouched
            Xh = np.stack([probs[max(0, min(1, int(t)))].x0 + probs[0].sigma0 * rng.normal(size=D) for t in tof_h]).astype(dtype)
            X, tof = torch.from_numpy(Xh).cuda(), torch.from_numpy(tof_h).cuda()
            ref = {"cost": torch.empty(n, dtype=tdt, device="cuda"), "grad": torch.empty(n, D, dtype=tdt, device="cuda"),
                   "minc": torch.empty(n, dtype=tdt, device="cuda"), "flags": torch.empty(n, dtype=torch.uint8, device="cuda")}
            got = {k: torch.full_like(v, 3) for k, v in ref.items()}
            rc = ctx.lib.mp_eval(ctx._h, n, vp(X.data_ptr()), vp(tof.data_ptr()), vp(ref["cost"].data_ptr()), None,
                                 vp(ref["grad"].data_ptr()), vp(ref["minc"].data_ptr()), vp(ref["flags"].data_ptr()), None,
                                 abi.MP_MEM_DEVICE)
            assert rc == 0, ctx.lib.mp_last_error(ctx._h)
            rc = ctx.lib.mp_eval_noalloc(ctx._h, n, vp(X.data_ptr()), vp(tof.data_ptr()), vp(got["cost"].data_ptr()),
                                         vp(got["grad"].data_ptr()), vp(got["minc"].data_ptr()), vp(got["flags"].data_ptr()))
            assert rc == 0, ctx.lib.mp_last_error(ctx._h)
            ctx.synchronize()
            bad = torch.zeros(n, dtype=torch.bool, device="cuda")
            bad[5] = bad[n - 3] = True
            assert torch.isnan(got["cost"][bad]).all() and torch.isnan(got["grad"][bad]).all() and (got["flags"][bad] == 0).all()
            assert torch.isfinite(got["cost"][~bad]).all()
            # mp_eval may take the fused kernel where mp_eval_noalloc always splits: the clearance profile (hence min clearance
            # and flags) is the same bits either way; cost and gradient are the same arithmetic unless only one of the two
            # launches takes the wide instance (S > 176: sums added in a different order)
            assert torch.equal(got["flags"], ref["flags"])
            assert torch.equal(got["minc"][~bad], ref["minc"][~bad])
            tol = 1e-12 if dtype is np.float64 else 1e-5
            c_g, c_r = got["cost"][~bad].double(), ref["cost"][~bad].double()
            assert float(((c_g - c_r).abs() / c_r.abs().clamp_min(1e-12)).max()) <= tol
            g_g, g_r = got["grad"][~bad].double(), ref["grad"][~bad].double()
            scale = g_r.abs().amax(dim=1, keepdim=True).clamp_min(H.GRAD_FLOOR)
            assert float(((g_g - g_r).abs() / scale).max()) <= (1e-11 if dtype is np.float64 else 1e-4)
        ctx.close()


def test_noalloc_calls_from_two_contexts_do_not_race():
    """Two contexts with streams of their own on one GPU (different problems, different batch sizes) issue their
    mp_eval_noalloc calls back to back without any host synchronisation in between: the library orders the users of the
    GPU's one scratch array on the device, so every call still gives mp_eval's bits (round-2 advisor finding: overlapping
    calls silently corrupted each other's tables)."""
    import torch

    from motion_planning_amd import _native as native
    from motion_planning_amd.synthetic import synthetic_candidates, synthetic_problem

    vp = C.c_void_p
    probs = [synthetic_problem(native.build_basis, 300, 256, seed=1, size_range=(0.05, 0.3)),
             synthetic_problem(native.build_basis, 50, 64, seed=0)]
    sizes = [3000, 20000]
    ctxs, Xs, refs, gots = [], [], [], []
    for p, n in zip(probs, sizes):
        ctx = native.Context(0)
        ctx.set_problem(p)
        X = torch.from_numpy(synthetic_candidates(p, n, seed=21)).cuda()
        ref = {"cost": torch.empty(n, dtype=torch.float64, device="cuda"), "grad": torch.empty(n, p.dim, dtype=torch.float64, device="cuda"),
               "flags": torch.empty(n, dtype=torch.uint8, device="cuda")}
        rc = ctx.lib.mp_eval(ctx._h, n, vp(X.data_ptr()), None, vp(ref["cost"].data_ptr()), None, vp(ref["grad"].data_ptr()), None,
                             vp(ref["flags"].data_ptr()), None, abi.MP_MEM_DEVICE)
        assert rc == 0, ctx.lib.mp_last_error(ctx._h)
        ctx.synchronize()
        ctxs.append(ctx); Xs.append(X); refs.append(ref)
        gots.append([{k: torch.zeros_like(v) for k, v in ref.items()} for _ in range(6)])
    for rep in range(6):          # A B A B ... on two streams, nothing waits on the host
        for ctx, X, got, n in zip(ctxs, Xs, gots, sizes):
            g = got[rep]
            rc = ctx.lib.mp_eval_noalloc(ctx._h, n, vp(X.data_ptr()), None, vp(g["cost"].data_ptr()), vp(g["grad"].data_ptr()), None,
                                         vp(g["flags"].data_ptr()))
            assert rc == 0, ctx.lib.mp_last_error(ctx._h)
    for ctx in ctxs:
        ctx.synchronize()
    for ref, got in zip(refs, gots):
        for g in got:
            for k in ref:
                assert torch.equal(g[k], ref[k]), k
    for ctx in ctxs:
        ctx.close()


def test_noalloc_limits_and_errors():
    import torch

    from motion_planning_amd import _native as native
    from motion_planning_amd.synthetic import synthetic_problem
    from tests import helpers as H

    prob = synthetic_problem(native.build_basis, 50, 64, seed=0)
    ctx = native.Context(0)
    ctx.set_problem(prob)
    vp = C.c_void_p
    per_cand = 64 * 12
    n_max = (MP_NOALLOC_SCRATCH_BYTES - 255) // per_cand
    n_over = MP_NOALLOC_SCRATCH_BYTES // per_cand + 1
    X = torch.zeros(n_over, prob.dim, dtype=torch.float64, device="cuda")
    X[:] = torch.from_numpy(prob.x0).cuda()
    cost = torch.empty(n_over, dtype=torch.float64, device="cuda")
    rc = ctx.lib.mp_eval_noalloc(ctx._h, n_over, vp(X.data_ptr()), None, vp(cost.data_ptr()), None, None, None)
    assert rc == abi.MP_E_ARG and b"MP_NOALLOC_SCRATCH_BYTES" in ctx.lib.mp_last_error(ctx._h)
    rc = ctx.lib.mp_eval_noalloc(ctx._h, n_max, vp(X.data_ptr()), None, vp(cost.data_ptr()), None, None, None)   # the largest batch that fits
    assert rc == 0, ctx.lib.mp_last_error(ctx._h)
    ctx.synchronize()
    assert torch.equal(cost[:n_max], cost[:1].expand(n_max)) and torch.isfinite(cost[0])
    assert ctx.lib.mp_eval_noalloc(ctx._h, 0, None, None, None, None, None, None) == 0
    assert ctx.lib.mp_eval_noalloc(ctx._h, 4, None, None, vp(cost.data_ptr()), None, None, None) == abi.MP_E_ARG
    # the point model has no scratch-using kernel
    case = next(c for c in H.load_cases() if c["name"] == "front_point")
    ctx.set_problem(H.problem_for_case(case))
    rc = ctx.lib.mp_eval_noalloc(ctx._h, 4, vp(X.data_ptr()), None, vp(cost.data_ptr()), None, None, None)
    assert rc == abi.MP_E_UNSUPPORTED
    ctx.close()


def test_noalloc_call_under_stream_capture():
    """What the entry point is for: a FIRST call of a shape inside a graph capture (mp_eval refuses there when its split
    tables would have to grow, tests/test_gpu_evidence.py::test_capture_refuses_to_allocate)."""
    import torch

    from motion_planning_amd import _native as native
    from motion_planning_amd.synthetic import synthetic_candidates, synthetic_problem

    prob = synthetic_problem(native.build_basis, 200, 64, seed=0)
    n = 2048
    ctx = native.Context(0)
    ctx.set_problem(prob)
    X = torch.from_numpy(synthetic_candidates(prob, n, seed=8)).cuda()
    cost = torch.zeros(n, dtype=torch.float64, device="cuda")
    grad = torch.zeros(n, prob.dim, dtype=torch.float64, device="cuda")
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.eval_device_noalloc(n, X.data_ptr(), cost.data_ptr(), grad.data_ptr(), 0, flags.data_ptr())
    ctx.set_stream(None)
    g.replay()
    torch.cuda.synchronize()
    ref = ctx.eval(X.cpu().numpy(), want_grad=True, want_terms=False)
    np.testing.assert_array_equal(cost.cpu().numpy(), ref["cost"])
    np.testing.assert_array_equal(grad.cpu().numpy(), ref["grad"])
    np.testing.assert_array_equal(flags.cpu().numpy(), ref["flags"])
    ctx.close()
