"""Batched, device-resident and multi-GPU evaluation.

The reference evaluates a CEM population with a Python list comprehension
(``objective_batch``, geom/spline_opt.py:301-302).  Here a batch is one kernel launch on
tensors that stay in HBM (``DeviceBatch``), and a batch larger than one GPU is sharded by
candidate index over the ranks of a ``torch.distributed`` job — one process per GPU, no
exchange on the data path, and ONE all-gather (RCCL over xGMI with the ``nccl`` backend) of
the per-candidate ``(cost, min_clearance, flags)`` at the end.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _abi as abi
from .problem import PathProblem


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Block partition by index: rank r owns candidates [n*r//world, n*(r+1)//world)."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


_TORCH_DT = {np.dtype(np.float64): torch.float64, np.dtype(np.float32): torch.float32}


class DeviceBatch:
    """One GPU, one problem (or several tasks), outputs preallocated in HBM.

    ``evaluate(X)`` launches on torch's current stream and returns views of the resident
    output tensors; nothing is copied to the host."""

    def __init__(self, problems: Sequence[PathProblem] | PathProblem, capacity: int, device: Optional[int] = None,
                 dtype=np.float64, want_grad: bool = True, want_terms: bool = False, cands_per_task: int = 0,
                 result_slots: int = 1):
        from . import _native

        if not torch.cuda.is_available():
            raise RuntimeError("DeviceBatch needs a GPU (there is no CPU fallback)")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.np_dtype = np.dtype(dtype)
        self.tdtype = _TORCH_DT[self.np_dtype]
        self.ctx = _native.Context(self.device, dtype)
        self.ctx.set_problem(problems)
        if cands_per_task:
            self.ctx.set_cands_per_task(cands_per_task)
        self.problem = self.ctx.problem
        self.capacity = int(capacity)
        self._stream = -1   # no stream bound yet (0 is a valid handle: the default stream)
        dev = torch.device("cuda", self.device)
        D = self.problem.dim
        # cost and min clearance are the two rows of one buffer, so that a job can all-gather both with one collective
        # straight from the kernel's output (no packing kernel in between)
        # ... and the flags follow in the same allocation: `results_bytes` (cost | min clearance | flags as raw bytes) is
        # what a job hands to ONE all-gather per step
        es = self.np_dtype.itemsize
        # `result_slots` > 1: a ring of result buffers in ONE allocation (evaluate(..., slot=k) writes ring entry k), so that
        # a job can hand the results of several consecutive batches to one collective (`results_ring[:k]`)
        self.result_slots = max(1, int(result_slots))
        self.slot_bytes = (2 * es + 1) * capacity
        self.results_ring = torch.empty(self.result_slots, self.slot_bytes, dtype=torch.uint8, device=dev)
        self._slot_views = []
        for k in range(self.result_slots):
            rb = self.results_ring[k]
            cm = rb[:2 * es * capacity].view(self.tdtype).view(2, capacity)
            self._slot_views.append((rb, cm, cm[0], cm[1], rb[2 * es * capacity:]))
        self.results_bytes, self.cost_minc, self.cost, self.min_clear, self.flags = self._slot_views[0]
        self.grad = torch.empty(capacity, D, dtype=self.tdtype, device=dev) if want_grad else None
        self.terms = torch.empty(capacity, abi.MP_NTERMS, dtype=self.tdtype, device=dev) if want_terms else None

    def evaluate(self, X: torch.Tensor, task_of: Optional[torch.Tensor] = None, slot: int = 0) -> Dict[str, torch.Tensor]:
        n = X.shape[0]
        if not 0 <= slot < self.result_slots:
            raise ValueError(f"result slot {slot} outside the ring of {self.result_slots}")
        _, _, cost, min_clear, flags = self._slot_views[slot]
        if n > self.capacity:
            raise ValueError("batch larger than the preallocated capacity")
        if X.dtype != self.tdtype or not X.is_cuda or not X.is_contiguous() or X.ndim != 2 or X.shape[1] != self.problem.dim:
            raise ValueError("X must be a contiguous CUDA tensor of the context dtype with D columns")
        if X.device.index != self.device:
            raise ValueError(f"X lives on cuda:{X.device.index}, this batch on cuda:{self.device}")
        if task_of is not None:
            # the library reads task_of as int32 device memory and does not range-check it on the host (out-of-range
            # entries come back as NaN cost / flags 0 from the kernel); torch's default index dtype is int64
            if (task_of.dtype != torch.int32 or not task_of.is_cuda or task_of.device.index != self.device
                    or not task_of.is_contiguous() or task_of.ndim != 1 or task_of.numel() < n):
                raise ValueError("task_of must be a contiguous 1-D int32 CUDA tensor on this batch's device with at least n entries")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if stream != self._stream:
            self.ctx.set_stream(stream)
            self._stream = stream
        self.ctx.eval_device(
            n, X.data_ptr(), cost.data_ptr(), self.terms.data_ptr() if self.terms is not None else 0,
            self.grad.data_ptr() if self.grad is not None else 0, min_clear.data_ptr(), flags.data_ptr(), 0,
            task_of.data_ptr() if task_of is not None else 0,
        )
        out = {"cost": cost[:n], "min_clear": min_clear[:n], "flags": flags[:n]}
        if self.grad is not None:
            out["grad"] = self.grad[:n]
        if self.terms is not None:
            out["terms"] = self.terms[:n]
        return out

    def close(self) -> None:
        self.ctx.close()


def pack_results(cost: torch.Tensor, min_clear: torch.Tensor, flags: torch.Tensor) -> torch.Tensor:
    """(n,) cost, (n,) min clearance, (n,) flags -> (n, 3) tensor of the cost dtype (flags are 0..3,
    exact in any float type) so that one collective moves everything."""
    return torch.stack([cost, min_clear, flags.to(cost.dtype)], dim=1).contiguous()


def allgather_results(local: torch.Tensor, n_total: int, rank: int, world: int) -> torch.Tensor:
    """The single collective of the path: every rank ends up with all (n_total, 3) rows.

    Shards differ by at most one row, so each rank pads to the largest shard for one
    ``all_gather_into_tensor`` and the padding is dropped afterwards."""
    import torch.distributed as dist

    if world == 1:
        return local
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    padded = torch.zeros(width, local.shape[1], dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    gathered = torch.empty(world * width, local.shape[1], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, padded)
    parts = [gathered[r * width: r * width + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)


def evaluate_sharded(evaluate_local: Callable[[torch.Tensor], Dict[str, torch.Tensor]], X_global: torch.Tensor,
                     rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Evaluate rows [lo, hi) of X_global on this rank and all-gather (cost, min_clear, flags).

    ``evaluate_local`` is ``DeviceBatch.evaluate`` on a GPU job; the CPU test tier passes an
    oracle-backed callable over the ``gloo`` backend."""
    n = X_global.shape[0]
    lo, hi = shard_bounds(n, rank, world)
    out = evaluate_local(X_global[lo:hi].contiguous())
    packed = pack_results(out["cost"], out["min_clear"], out["flags"])
    full = allgather_results(packed, n, rank, world)
    return {"cost": full[:, 0], "min_clear": full[:, 1], "flags": full[:, 2].to(torch.uint8),
            "local": out, "bounds": (lo, hi)}


def best_feasible(cost: torch.Tensor, flags: torch.Tensor) -> Tuple[int, float, bool]:
    """Index and cost of the best candidate, preferring feasible ones (best-of-restarts)."""
    feas = (flags & abi.MP_FLAG_FEASIBLE) != 0
    if bool(feas.any()):
        masked = torch.where(feas, cost, torch.full_like(cost, float("inf")))
        i = int(torch.argmin(masked))
        return i, float(cost[i]), True
    i = int(torch.argmin(cost))
    return i, float(cost[i]), False


def solve_scenarios_cem(problems: Sequence[PathProblem], restarts: int = 32, device: int = 0, dtype=np.float64,
                        restart_seed: int = 0, scenario_base: int = 0, **cem_options) -> Dict[str, np.ndarray]:
    """Scenario-parallel planning (BASELINE config 5): every scenario gets ``restarts`` independent CEM
    solves started from x0 + sigma0 * eps (restart 0 from x0 itself), all instances advance in lock-step
    on the GPU, and the best restart of each scenario is returned.

    "solved" follows SURVEY.md §8d: the best candidate is feasible (min clearance >= required).
    ``scenario_base``: global index of ``problems[0]`` when the list is one rank's contiguous share of a longer one.  The
    restart offsets and the device noise streams are keyed by GLOBAL scenario / instance index, so a scenario's result
    does not depend on how the list is partitioned over ranks (tests/test_gpu_evidence.py)."""
    from . import _native

    p0 = problems[0]
    n_s, D = len(problems), p0.dim
    x0 = np.empty((n_s, restarts, D))
    s0 = np.empty((n_s, restarts, D))
    for i, p in enumerate(problems):
        eps = np.random.default_rng([int(restart_seed), int(scenario_base) + i]).normal(size=(restarts, D))
        eps[0] = 0.0
        x0[i] = p.x0[None, :] + eps * p.sigma0[None, :]
        s0[i] = p.sigma0[None, :]
    ctx = _native.Context(device, dtype)
    try:
        ctx.set_problem(list(problems))
        task_of = np.repeat(np.arange(n_s, dtype=np.int32), restarts)
        res = ctx.cem_solve(x0.reshape(-1, D), s0.reshape(-1, D), task_of=task_of,
                            instance_base=int(scenario_base) * int(restarts), **cem_options)
    finally:
        ctx.close()
    fun = res["fun"].astype(float).reshape(n_s, restarts)
    flags = res["flags"].reshape(n_s, restarts)
    feas = (flags & abi.MP_FLAG_FEASIBLE) != 0
    masked = np.where(feas, fun, np.inf)
    pick = np.where(feas.any(axis=1), np.argmin(masked, axis=1), np.argmin(fun, axis=1))
    rows = np.arange(n_s)
    return {
        "x": res["x"].reshape(n_s, restarts, D)[rows, pick],
        "fun": fun[rows, pick],
        "min_clear": res["min_clear"].astype(float).reshape(n_s, restarts)[rows, pick],
        "solved": feas[rows, pick],
        "restart": pick,
        "nit": res["nit"].reshape(n_s, restarts),
        "all_fun": fun,
    }


def refine_gradient(db: DeviceBatch, X: torch.Tensor, iters: int = 20, step0: float = 0.05,
                    scale: Optional[torch.Tensor] = None, chunk: int = 2048, **_ignored) -> Dict[str, torch.Tensor]:
    """Batched local refinement with the analytic gradient, device resident (the reference has no counterpart: its
    local stage is the derivative-free Powell search, geom/spline_opt.py:339-344).  Same algorithm as
    ``spline_opt._refine_with_gradient``, on tensors that stay in HBM: per iteration every candidate evaluates a fan of
    trial points — line searches along the gradient and along a BFGS direction, multi-scale probes along every
    coordinate, repeats of its last move — and moves to the best one if that lowers its cost (monotone).
    ``scale`` (D,) preconditions the mixed units of x (metres and degrees); default: the problem's sigma0.
    One ``mp_eval`` launch per iteration and ``chunk`` candidates (a fan is ~110 trial points per candidate)."""
    from .spline_opt import _COMPASS_LEVELS, _LINE_ALPHAS, _PATTERN, _QN_ALPHAS

    if db.grad is None:
        raise ValueError("DeviceBatch was created without want_grad")
    dev, dt = X.device, X.dtype
    if scale is None:
        scale = torch.as_tensor(db.problem.sigma0, dtype=dt, device=dev)
    n, D = X.shape
    al = torch.tensor(_LINE_ALPHAS, dtype=dt, device=dev)
    qa = torch.tensor(_QN_ALPHAS, dtype=dt, device=dev)
    ga = torch.tensor(_PATTERN, dtype=dt, device=dev)
    A = al.numel()
    probes = torch.zeros(2 * D * len(_COMPASS_LEVELS), D, dtype=dt, device=dev)
    row = 0
    for level in _COMPASS_LEVELS:
        for j in range(D):
            probes[row, j], probes[row + 1, j] = level * scale[j], -level * scale[j]
            row += 2
    K = A + qa.numel() + probes.shape[0] + ga.numel()
    chunk = max(1, min(int(chunk), db.capacity // K))
    if chunk < 1 or db.capacity < K:
        raise ValueError(f"DeviceBatch capacity {db.capacity} is below one fan of {K} trial points")
    x = X.clone()
    out = {"x": x, "cost": torch.empty(n, dtype=dt, device=dev), "min_clear": torch.empty(n, dtype=dt, device=dev),
           "flags": torch.empty(n, dtype=torch.uint8, device=dev)}
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        R = hi - lo
        xc = x[lo:hi]
        o = db.evaluate(xc.contiguous())
        cost, grad = o["cost"].clone(), o["grad"].clone()
        minc, flags = o["min_clear"].clone(), o["flags"].clone()
        step = torch.full((R,), float(step0), dtype=dt, device=dev)
        last = torch.zeros(R, D, dtype=dt, device=dev)
        Hinv = torch.zeros(R, D, D, dtype=dt, device=dev)
        started = torch.zeros(R, dtype=torch.bool, device=dev)
        eye = torch.eye(D, dtype=dt, device=dev)
        rows = torch.arange(R, device=dev)
        for _ in range(int(iters)):
            gz = grad * scale[None, :]
            unit = gz / gz.norm(dim=1, keepdim=True).clamp_min(1e-30)
            dz = -torch.einsum("rij,rj->ri", Hinv, gz)
            trials = torch.cat([
                xc[:, None, :] - (step[:, None, None] * al[None, :, None]) * (scale[None, :] * unit)[:, None, :],
                xc[:, None, :] + qa[None, :, None] * (dz * scale[None, :])[:, None, :],
                xc[:, None, :] + probes[None, :, :],
                xc[:, None, :] + ga[None, :, None] * last[:, None, :],
            ], dim=1)
            t = db.evaluate(trials.reshape(R * K, D).contiguous())
            c2 = t["cost"].reshape(R, K)
            c2 = torch.where(torch.isfinite(c2), c2, torch.full_like(c2, float("inf")))
            j = torch.argmin(c2, dim=1)
            best = c2[rows, j]
            better = best < cost
            pick = rows * K + j
            new_x = torch.where(better[:, None], trials[rows, j], xc)
            new_g = torch.where(better[:, None], t["grad"][pick], grad)
            minc = torch.where(better, t["min_clear"][pick], minc)
            flags = torch.where(better, t["flags"][pick], flags)
            sv, yv = (new_x - xc) / scale[None, :], (new_g - grad) * scale[None, :]
            sy = (sv * yv).sum(dim=1)
            ok = better & (sy > 1e-12 * sv.norm(dim=1) * yv.norm(dim=1))
            sy_safe = torch.where(ok, sy, torch.ones_like(sy))
            first = ok & ~started
            Hinv = torch.where(first[:, None, None], eye[None] * (sy_safe / (yv * yv).sum(dim=1).clamp_min(1e-300))[:, None, None], Hinv)
            started = started | first
            V = eye[None] - sv[:, :, None] * yv[:, None, :] / sy_safe[:, None, None]
            Hnew = V @ Hinv @ V.transpose(1, 2) + sv[:, :, None] * sv[:, None, :] / sy_safe[:, None, None]
            Hinv = torch.where(ok[:, None, None], Hnew, Hinv)
            last = torch.where(better[:, None], new_x - xc, torch.zeros_like(xc))
            xc, grad = new_x, new_g
            cost = torch.where(better, best, cost)
            on_line = better & (j < A)
            grow = on_line & (al[j.clamp_max(A - 1)] >= 2.0)
            step = torch.where(grow, step * 2.0, torch.where(on_line, step, step * 0.5)).clamp_min(1e-5)
        x[lo:hi] = xc
        out["cost"][lo:hi], out["min_clear"][lo:hi], out["flags"][lo:hi] = cost, minc, flags
    return out
