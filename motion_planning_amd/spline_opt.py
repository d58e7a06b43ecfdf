"""Planner / optimiser call surface of the reference (geom/spline_opt.py), backed by the
MI355X kernels.

``optimize_bspline_path`` (ref :391-746), ``optimize_bspline_with_vias`` (:749-849) and
``optimize_bspline_two_vias`` (:852-884) keep the reference's signatures, defaults, ValueErrors,
return tuples and ``info`` keys.  What changed is underneath: the objective closure is replaced
by one native call per optimiser query (`mp_eval` through ``NativeEvaluator``) — one candidate
for SciPy's Powell / Nelder-Mead, a whole population for the CEM — instead of S x B Python-level
FCL calls per evaluation.

The optimiser drivers stay on the host in "parity mode": SciPy's ``minimize`` and a CEM that
draws its noise from ``np.random.default_rng(seed)`` exactly like the reference, so that a
seeded solve visits the same candidates.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

try:
    from scipy.interpolate import make_interp_spline
    from scipy.optimize import minimize
except ImportError as e:  # same behaviour as the reference module (:4-8)
    raise ImportError("Please install SciPy: pip install scipy") from e

from . import _abi as abi
from .problem import PathProblem, build_path_problem, build_vias_problem, default_via_initialization

# ---------------------------------------------------------------------------------------------
# small host-side helpers that are part of the reference module's public surface
# ---------------------------------------------------------------------------------------------


def build_cubic_bspline(points: np.ndarray) -> Callable[[np.ndarray], np.ndarray]:
    """Interpolating cubic B-spline through N >= 4 waypoints, parameter u in [0, 1] (ref :11-22)."""
    n = points.shape[0]
    if n < 4:
        raise ValueError("Cubic B-spline interpolation requires at least 4 waypoints.")
    spl = make_interp_spline(np.linspace(0.0, 1.0, n), points, k=3, axis=0)
    return lambda uq: spl(np.asarray(uq, dtype=float))


def build_scalar_bspline(values: np.ndarray) -> Callable[[np.ndarray], np.ndarray]:
    """Scalar interpolating spline of degree min(3, n - 1) (ref :25-34)."""
    y = np.asarray(values, dtype=float).reshape(-1)
    if y.size < 2:
        raise ValueError("Scalar spline requires at least 2 control points.")
    spl = make_interp_spline(np.linspace(0.0, 1.0, y.size), y, k=min(3, y.size - 1), axis=0)
    return lambda uq: np.asarray(spl(np.asarray(uq, dtype=float)), dtype=float)


def yaw_deg_to_quat(yaw_deg: float) -> Tuple[float, float, float, float]:
    """[x, y, z, w] of a rotation by ``yaw_deg`` about +z (ref :37-40)."""
    half = 0.5 * np.deg2rad(float(yaw_deg))
    return (0.0, 0.0, float(np.sin(half)), float(np.cos(half)))


def sample_curve(S: Callable, n: int = 101) -> Tuple[np.ndarray, np.ndarray]:
    P = S(np.linspace(0.0, 1.0, n))
    return P, np.diff(P, axis=0)


def path_length(P: np.ndarray) -> float:
    return float(np.sum(np.linalg.norm(np.diff(P, axis=0), axis=1)))


def mean_turn_angle_deg(P: np.ndarray, eps: float = 1e-12) -> float:
    """Mean angle between consecutive polyline segments, degrees (ref :82-97)."""
    seg = np.diff(P, axis=0)
    if seg.shape[0] < 2:
        return 0.0
    a, b = seg[:-1], seg[1:]
    na, nb = np.linalg.norm(a, axis=1), np.linalg.norm(b, axis=1)
    ok = (na > eps) & (nb > eps)
    if not np.any(ok):
        return 0.0
    cosang = np.clip(np.sum(a[ok] * b[ok], axis=1) / (na[ok] * nb[ok]), -1.0, 1.0)
    return float(np.degrees(np.mean(np.arccos(cosang))))


def curvature_cost(P: np.ndarray) -> float:
    """Discrete bending energy sum(kappa^2 |P'|) du with two np.gradient passes (ref :63-79); 0 for fewer than 3 samples.
    Host helper for callers of the reference module; the kernels compute the same term on the device."""
    pts = np.asarray(P, dtype=float)
    m = pts.shape[0]
    if m < 3:
        return 0.0
    h = 1.0 / float(m - 1)
    vel = np.gradient(pts, h, axis=0)
    acc = np.gradient(vel, h, axis=0)
    v = np.linalg.norm(vel, axis=1)
    kappa = np.linalg.norm(np.cross(vel, acc), axis=1) / np.maximum(v, 1e-9) ** 3
    return float(np.sum(kappa * kappa * v) * h)


def yaw_smoothness_cost(yaw_deg_samples: np.ndarray) -> float:
    """Sum of squared second differences of the yaw profile, degrees^2 (ref :100-106)."""
    y = np.asarray(yaw_deg_samples, dtype=float).ravel()
    if y.size < 3:
        return 0.0
    return float(np.sum(np.diff(y, n=2) ** 2))


def _normalize(v: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """v / |v|, or zeros when |v| < eps (ref :109-114)."""
    flat = np.asarray(v, dtype=float).ravel()
    length = float(np.linalg.norm(flat))
    return flat / length if length >= eps else np.zeros_like(flat)


def _goal_approach_alignment_cost(P: np.ndarray, goal_normals: np.ndarray, terminal_fraction: float = 0.1) -> float:
    """(1 - clip(v . -s, -1, 1))^2 with v the direction of the last max(3, ceil(fraction * S)) samples and s the
    normalised resultant of the unit goal normals, first normal as fallback (ref :117-145)."""
    pts = np.asarray(P, dtype=float)
    normals = np.asarray(goal_normals, dtype=float)
    if pts.shape[0] < 3 or normals.size == 0:
        return 0.0
    window = max(3, int(np.ceil(float(terminal_fraction) * pts.shape[0])))
    tail = pts[-window:]
    if tail.shape[0] < 2:
        return 0.0
    heading = _normalize(tail[-1] - tail[0])          # the sum of the tail's differences telescopes
    units = np.array([_normalize(row) for row in normals.reshape(-1, 3)], dtype=float)
    outward = _normalize(units.sum(axis=0))
    if not outward.any():
        outward = _normalize(units[0])
    if not heading.any() or not outward.any():
        return 0.0
    return float((1.0 - np.clip(float(heading @ -outward), -1.0, 1.0)) ** 2)


def _path_distances(scene, P: np.ndarray, moving_block_size: Optional[Tuple[float, float, float]] = None,
                    moving_block_quat: Tuple[float, float, float, float] = (0.0, 0.0, 0.0, 1.0),
                    moving_block_quats: Optional[np.ndarray] = None, ignore_ids: Optional[List[str]] = None) -> np.ndarray:
    """Clearance profile of points or of a moving block along P (ref :148-186): point model without a block size,
    otherwise per-sample quaternions (shape-checked) or one fixed quaternion.  A scene that offers the batched queries
    of ``motion_planning_amd.scene.Scene`` answers the whole profile in one kernel launch; any other duck-typed scene
    (geom/scene.py:50, :70) is asked sample by sample like the reference does."""
    from .utils import quat_is_yaw_only, yaw_quat_to_deg

    pts = np.asarray(P, dtype=float).reshape(-1, 3)
    if moving_block_size is None:
        many = getattr(scene, "signed_distance_many", None)
        if callable(many):
            return np.asarray(many(pts), dtype=float)
        return np.array([scene.signed_distance(p) for p in pts], dtype=float)
    if moving_block_quats is not None:
        quats = np.asarray(moving_block_quats, dtype=float)
        if quats.shape != (pts.shape[0], 4):
            raise ValueError("moving_block_quats must have shape (len(P), 4)")
    else:
        quats = np.broadcast_to(np.asarray(moving_block_quat, dtype=float), (pts.shape[0], 4))
    many = getattr(scene, "signed_distance_block_many", None)
    if callable(many) and all(quat_is_yaw_only(q) for q in quats):
        yaw = np.array([yaw_quat_to_deg(q) for q in quats], dtype=float)
        return np.asarray(many(moving_block_size, pts, yaw, ignore_ids), dtype=float)
    return np.array([scene.signed_distance_block(size=moving_block_size, position=p, quat=tuple(q.tolist()), ignore_ids=ignore_ids)
                     for p, q in zip(pts, quats)], dtype=float)


def safety_cost(scene, P: np.ndarray, required_clearance: float, moving_block_size: Optional[Tuple[float, float, float]] = None,
                moving_block_quat: Tuple[float, float, float, float] = (0.0, 0.0, 0.0, 1.0),
                moving_block_quats: Optional[np.ndarray] = None, ignore_ids: Optional[List[str]] = None) -> float:
    """sum(max(0, required_clearance - d_i)^2) over the clearance profile (ref :189-211)."""
    d = _path_distances(scene, P, moving_block_size=moving_block_size, moving_block_quat=moving_block_quat,
                        moving_block_quats=moving_block_quats, ignore_ids=ignore_ids)
    short = np.maximum(0.0, float(required_clearance) - d)
    return float(short @ short)


_default_via_initialization = default_via_initialization

# ---------------------------------------------------------------------------------------------
# native evaluator
# ---------------------------------------------------------------------------------------------


class NativeEvaluator:
    """objective_batch of the reference (:301-302) as one kernel launch per call."""

    def __init__(self, problem: PathProblem, device: int = 0, dtype=np.float64):
        from . import _native

        self.problem = problem
        self.ctx = _native.Context(device, dtype)
        self.ctx.set_problem(problem)
        self.n_calls = 0
        self.n_evals = 0

    def __call__(self, X: np.ndarray, want_terms: bool = False, want_grad: bool = False, want_dist: bool = False):
        X = np.asarray(X, dtype=float)
        if X.ndim == 1:
            X = X[None, :]
        self.n_calls += 1
        self.n_evals += X.shape[0]
        out = self.ctx.eval(X, want_terms=want_terms, want_grad=want_grad, want_dist=want_dist)
        return {k: (None if v is None else np.asarray(v, dtype=float) if k != "flags" else v) for k, v in out.items()}

    def cem_multi(self, x0, sigma0, seeds, population_size: int = 48, elite_frac: float = 0.2, max_iter: int = 60,
                  tol: float = 1e-3, alpha: float = 0.7, min_sigma: float = 1e-3) -> List[Dict]:
        """``len(seeds)`` CEM runs (ref :222-277) from the same start, resident on the device (``mp_cem_solve``: sampling,
        evaluation, elite statistics and the stop rule of every run in lock-step, no host round trip per generation).  The
        noise comes from the device generator keyed by ``seeds[0]`` and the run index — not NumPy's stream: the runs follow
        the reference's update and stop rule, not its random numbers."""
        R = len(seeds)
        x0 = np.asarray(x0, dtype=float)
        out = self.ctx.cem_solve(np.repeat(x0[None, :], R, axis=0), np.asarray(sigma0, dtype=float), population_size=population_size,
                                 elite_frac=elite_frac, max_iter=max_iter, tol=tol, alpha=alpha, min_sigma=min_sigma, seed=seeds[0])
        nit = np.asarray(out["nit"], dtype=int)
        self.n_calls += int(nit.max())                       # one evaluation launch per generation for all runs
        self.n_evals += int(nit.sum()) * int(population_size)
        return [{"x": np.asarray(out["x"][k], dtype=float), "fun": float(out["fun"][k]), "nit": int(nit[k]), "success": True,
                 "message": "CEM finished"} for k in range(R)]

    def clearance_profile(self, P: np.ndarray, yaw_deg: Optional[np.ndarray]) -> np.ndarray:
        """_path_distances over arbitrary samples (ref :148-186) with the problem's scene."""
        p = self.problem
        if p.boxes.shape[0] == 0:
            return np.full(P.shape[0], np.inf)
        if p.model == abi.MP_MODEL_BOX:
            yaw = np.zeros(P.shape[0]) if yaw_deg is None else yaw_deg
            return np.asarray(self.ctx.box_sdf(2.0 * p.half_extents, P, yaw), dtype=float)
        return np.asarray(self.ctx.point_sdf(P, p.point_radius), dtype=float)

    def close(self) -> None:
        self.ctx.close()


EvaluatorFactory = Callable[[PathProblem], Callable]

# ---------------------------------------------------------------------------------------------
# optimiser drivers (host, parity mode)
# ---------------------------------------------------------------------------------------------


def _cem_optimize(objective, x0, sigma0, population_size: int = 96, elite_frac: float = 0.2, max_iter: int = 120,
                  tol: float = 1e-3, alpha: float = 0.7, min_sigma: float = 1e-3, seed: Optional[int] = None):
    """Cross-entropy method with the reference's update and stopping rule (ref :222-277).

    ``objective`` maps a (population, D) array to (population,) costs: one native launch."""
    rng = np.random.default_rng(seed)
    mean = x0.astype(float).copy()
    spread = np.maximum(np.asarray(sigma0, dtype=float), min_sigma)
    dim = mean.size
    n_elite = max(2, int(np.ceil(population_size * elite_frac)))
    incumbent, incumbent_val, previous_val = mean.copy(), float("inf"), float("inf")
    generations = 0
    while generations < max_iter:
        generations += 1
        noise = rng.normal(size=(population_size, dim))
        population = mean[None, :] + noise * spread[None, :]
        values = objective(population)
        lead = int(np.argmin(values))
        if values[lead] < incumbent_val:
            incumbent_val = float(values[lead])
            incumbent = population[lead].copy()
        chosen = population[np.argpartition(values, n_elite - 1)[:n_elite]]
        mean = alpha * mean + (1.0 - alpha) * np.mean(chosen, axis=0)
        spread = np.maximum(alpha * spread + (1.0 - alpha) * (np.std(chosen, axis=0) + min_sigma), min_sigma)
        if np.abs(previous_val - incumbent_val) < tol and float(np.max(spread)) < 0.03:
            break
        previous_val = incumbent_val
    return {"x": incumbent, "fun": incumbent_val, "nit": generations, "success": True, "message": "CEM finished"}


_CEM_DEFAULTS = dict(population_size=64, elite_frac=0.2, max_iter=90, tol=1e-3, alpha=0.7, min_sigma=1e-3, seed=None)
_HYBRID_CEM_DEFAULTS = dict(population_size=48, elite_frac=0.2, max_iter=60, tol=1e-3, alpha=0.7, min_sigma=1e-3, seed=None)
_HYBRID_POWELL_DEFAULTS = dict(maxiter=80, xtol=1e-3, ftol=1e-3)


_GRAD_DEFAULTS = dict(iters=40, n_starts=8, copies=2, jitter=0.05, step0=0.05)
_LINE_ALPHAS = (0.125, 0.25, 0.5, 1.0, 2.0, 4.0, 8.0)        # gradient line search: multiples of the adaptive step
_QN_ALPHAS = (0.03, 0.1, 0.3, 0.6, 1.0, 1.5, 2.5)            # quasi-Newton line search: multiples of the BFGS step
_COMPASS_LEVELS = (0.003, 0.01, 0.03, 0.1, 0.3, 1.0)         # coordinate probes: +- level * sigma0_j
_PATTERN = (1.0, 2.0)                                        # repeats of the last accepted move


def _cem_optimize_multi(objective, x0, sigma0, seeds, population_size: int = 48, elite_frac: float = 0.2, max_iter: int = 60,
                        tol: float = 1e-3, alpha: float = 0.7, min_sigma: float = 1e-3) -> List[Dict]:
    """``len(seeds)`` independent runs of ``_cem_optimize`` (ref :222-277) advanced in lock-step: every generation the
    populations of the runs still going are evaluated by ONE ``objective`` call.  Run k is, bit for bit, what
    ``_cem_optimize(..., seed=seeds[k])`` returns (same generator, same draw order, same update); only the number of
    evaluator calls changes — max over the runs instead of their sum."""
    R = len(seeds)
    rngs = [np.random.default_rng(sd) for sd in seeds]
    x0 = np.asarray(x0, dtype=float)
    dim = x0.size
    mean = np.repeat(x0[None, :], R, axis=0)
    spread = np.repeat(np.maximum(np.asarray(sigma0, dtype=float), min_sigma)[None, :], R, axis=0)
    best_x, best = mean.copy(), np.full(R, np.inf)
    prev = np.full(R, np.inf)
    nit = np.zeros(R, dtype=int)
    live = np.ones(R, dtype=bool)
    n_elite = max(2, int(np.ceil(population_size * elite_frac)))
    for _ in range(int(max_iter)):
        idx = np.nonzero(live)[0]
        if idx.size == 0:
            break
        nit[idx] += 1
        # the per-run statistics below are the reference's expressions with one leading axis (the run): NumPy reduces
        # each run's rows in the same order as the single-run code, so the bits are the same (tests/test_host_logic.py)
        noise = np.stack([rngs[k].normal(size=(population_size, dim)) for k in idx])
        pops = mean[idx, None, :] + noise * spread[idx, None, :]
        values = np.asarray(objective(pops.reshape(-1, dim)), dtype=float).reshape(idx.size, population_size)
        rows = np.arange(idx.size)
        lead = np.argmin(values, axis=1)
        lead_val = values[rows, lead]
        better = lead_val < best[idx]
        best[idx[better]] = lead_val[better]
        best_x[idx[better]] = pops[rows[better], lead[better]]
        elite = np.argpartition(values, n_elite - 1, axis=1)[:, :n_elite]
        chosen = np.take_along_axis(pops, elite[:, :, None], axis=1)
        mean[idx] = alpha * mean[idx] + (1.0 - alpha) * np.mean(chosen, axis=1)
        spread[idx] = np.maximum(alpha * spread[idx] + (1.0 - alpha) * (np.std(chosen, axis=1) + min_sigma), min_sigma)
        stop = (np.abs(prev[idx] - best[idx]) < tol) & (np.max(spread[idx], axis=1) < 0.03)
        live[idx[stop]] = False
        prev[idx] = best[idx]
    return [{"x": best_x[k].copy(), "fun": float(best[k]), "nit": int(nit[k]), "success": True, "message": "CEM finished"}
            for k in range(R)]


def _refine_with_gradient(value_and_grad: Callable, x: np.ndarray, sigma0: np.ndarray, iters: int = 40, copies: int = 2,
                          jitter: float = 0.05, step0: float = 0.05, seed: Optional[int] = None, ftol: float = 1e-10,
                          patience: int = 3, **_ignored) -> Dict:
    """Batched local refinement with the analytic (sub)gradient — the stand-in for the reference's derivative-free Powell
    stage (ref :338-359) that a batched evaluator makes cheap.  ``x``: one start (D,) or several (R, D); every start is
    refined ``copies`` times (the first copy exact, the others jittered by ``jitter * sigma0``), all of them together, ONE
    evaluator call per iteration.

    Per iteration and copy a fan of trial points is evaluated and the best one is taken if it lowers the cost:
    a line search along the gradient (normalised in sigma0 units: metres and degrees mix in x) at seven multiples of an
    adaptive step; the same along the BFGS quasi-Newton direction built from the accepted moves (follows curved valleys
    that defeat the plain gradient); +-sigma0_j * level probes along every coordinate at six scales (the batched
    counterpart of the bracketing line searches Powell runs along each direction: robust at the kinks of the
    sub-gradient); and repeats of the last accepted move.  Monotone: a copy never gets worse.
    Stops before ``iters`` once the best cost has not improved by more than ``ftol`` (relative) for ``patience``
    consecutive iterations.  ``value_and_grad(X) -> (cost (n,), grad (n, D))``."""
    rng = np.random.default_rng(seed)
    scale = np.asarray(sigma0, dtype=float)
    starts = np.atleast_2d(np.asarray(x, dtype=float))
    D = starts.shape[1]
    X = np.repeat(starts, max(1, int(copies)), axis=0)
    extra = np.ones(X.shape[0], dtype=bool)
    extra[:: max(1, int(copies))] = False
    X[extra] += jitter * scale[None, :] * rng.normal(size=(int(extra.sum()), D))
    R = X.shape[0]
    cost, grad = value_and_grad(X)
    cost, grad = np.array(cost, dtype=float), np.array(grad, dtype=float)
    n_eval = R
    al, qa, ga = np.asarray(_LINE_ALPHAS), np.asarray(_QN_ALPHAS), np.asarray(_PATTERN)
    A = al.size
    probes = np.zeros((2 * D * len(_COMPASS_LEVELS), D))
    row = 0
    for level in _COMPASS_LEVELS:
        for j in range(D):
            probes[row, j], probes[row + 1, j] = level * scale[j], -level * scale[j]
            row += 2
    step = np.full(R, float(step0))
    last = np.zeros((R, D))
    Hinv = np.zeros((R, D, D))          # inverse Hessian estimate in scaled coordinates z = x / sigma0; zero until the first update
    rows = np.arange(R)
    eye = np.eye(D)
    Q, G = qa.size, ga.size
    K = A + Q + probes.shape[0] + G
    trials = np.empty((R, K, D))
    nit, stalled, best_so_far = 0, 0, float(np.min(cost))
    for _ in range(int(iters)):
        nit += 1
        gz = grad * scale[None, :]
        unit = gz / np.maximum(np.linalg.norm(gz, axis=1, keepdims=True), 1e-30)
        dz = -np.einsum("rij,rj->ri", Hinv, gz)
        # the fan, written into one buffer (same expressions as X - (step al) (scale unit), X + qa (dz scale), X + probes,
        # X + ga last; no concatenation, no temporaries of the fan's size)
        Xb = X[:, None, :]
        part = trials[:, :A]
        np.multiply(step[:, None, None] * al[None, :, None], (scale[None, :] * unit)[:, None, :], out=part)
        np.subtract(Xb, part, out=part)
        part = trials[:, A:A + Q]
        np.multiply(qa[None, :, None], (dz * scale[None, :])[:, None, :], out=part)
        np.add(Xb, part, out=part)
        np.add(Xb, probes[None, :, :], out=trials[:, A + Q:K - G])
        part = trials[:, K - G:]
        np.multiply(ga[None, :, None], last[:, None, :], out=part)
        np.add(Xb, part, out=part)
        c2, g2 = value_and_grad(trials.reshape(R * K, D))
        n_eval += R * K
        c2 = np.asarray(c2, dtype=float).reshape(R, K)
        g2 = np.asarray(g2, dtype=float).reshape(R, K, D)
        c2 = np.where(np.isfinite(c2), c2, np.inf)
        j = np.argmin(c2, axis=1)
        best = c2[rows, j]
        better = best < cost
        new_x = np.where(better[:, None], trials[rows, j], X)
        new_g = np.where(better[:, None], g2[rows, j], grad)
        # BFGS update of the accepted moves that satisfy the curvature condition
        sv, yv = (new_x - X) / scale[None, :], (new_g - grad) * scale[None, :]
        sy = np.sum(sv * yv, axis=1)
        ok = better & (sy > 1e-12 * np.linalg.norm(sv, axis=1) * np.linalg.norm(yv, axis=1))
        if ok.any():   # all accepted copies at once (stacked D x D products)
            u = np.nonzero(ok)[0]
            s_u, y_u, sy_u = sv[u], yv[u], sy[u][:, None, None]
            H = Hinv[u]
            fresh = ~H.reshape(u.size, -1).any(axis=1)
            if fresh.any():
                H[fresh] = eye[None, :, :] * (sy[u][fresh] / np.sum(y_u[fresh] * y_u[fresh], axis=1))[:, None, None]
            V = eye[None, :, :] - s_u[:, :, None] * y_u[:, None, :] / sy_u
            Hinv[u] = V @ H @ V.transpose(0, 2, 1) + s_u[:, :, None] * s_u[:, None, :] / sy_u
        last = np.where(better[:, None], new_x - X, 0.0)
        X, grad = new_x, new_g
        cost = np.where(better, best, cost)
        on_line = better & (j < A)
        grow = on_line & (al[np.minimum(j, A - 1)] >= 2.0)
        step = np.maximum(np.where(grow, step * 2.0, np.where(on_line, step, step * 0.5)), 1e-5)
        now = float(np.min(cost))
        stalled = stalled + 1 if best_so_far - now <= ftol * abs(now) else 0
        best_so_far = min(best_so_far, now)
        if stalled >= int(patience):
            break
    i = int(np.argmin(cost))
    return {"x": X[i].copy(), "fun": float(cost[i]), "nit": int(nit), "n_eval": int(n_eval), "all_fun": cost.copy()}


def _solve_optimizer(objective_single, x0, sigma0, method: str, options: Optional[Dict] = None,
                     objective_batch: Optional[Callable] = None, value_and_grad: Optional[Callable] = None,
                     cem_multi: Optional[Callable] = None) -> Dict:
    """Method dispatch and option merging of the reference (ref :280-388).

    ``objective_single(x)`` returns a tuple whose element 0 is the cost; ``objective_batch(X)``
    (optional) returns the cost vector of a population in one call.  One method beyond the reference's:
    ``"CEM-GRAD"`` — the hybrid with the Powell stage replaced by ``_refine_with_gradient`` (needs
    ``value_and_grad``; options ``{"cem": {...}, "grad": {...}, "device_cem": bool}``; ``cem_multi``: the evaluator's
    device-resident lock-step CEM, used for stage 1 unless ``device_cem`` is False — then, and with an evaluator that has
    none, the runs are NumPy-seeded host loops and run 0 is the reference's own seeded run)."""
    if objective_batch is None:
        def objective_batch(X):
            return np.array([objective_single(x)[0] for x in X], dtype=float)

    def scalar(x):
        return objective_single(x)[0]

    name = method.upper()
    if name == "CEM":
        cem = dict(_CEM_DEFAULTS)
        if options:
            cem.update(options)
        res = _cem_optimize(objective_batch, x0=x0, sigma0=sigma0, **cem)
        return {"x": res["x"], "success": bool(res["success"]), "message": str(res["message"]), "nit": int(res["nit"]),
                "fun": float(res["fun"])}

    if name == "CEM-GRAD":
        if value_and_grad is None:
            raise ValueError("method 'CEM-GRAD' needs an evaluator with gradients")
        cem, grad_opt = dict(_HYBRID_CEM_DEFAULTS), dict(_GRAD_DEFAULTS)
        device_cem = bool(options.get("device_cem", True)) if options else True   # when the evaluator offers it (cem_multi)
        if options:
            if isinstance(options.get("cem"), dict):
                cem.update(options["cem"])
            if isinstance(options.get("grad"), dict):
                grad_opt.update(options["grad"])
            for key, val in options.items():
                if key in cem:
                    cem[key] = val
                if key in grad_opt:
                    grad_opt[key] = val
        if "restarts" in grad_opt:      # round-1 option name: total number of refined copies
            grad_opt["copies"] = max(1, int(grad_opt.pop("restarts")) // max(1, int(grad_opt["n_starts"])))
        # several CEM runs in lock-step (run 0 is the reference's seeded run), every run's best refined together
        seed = cem.pop("seed")
        n_starts = max(1, int(grad_opt.pop("n_starts")))
        seeds = [seed if k == 0 else (None if seed is None else int(seed) + 1000 * k) for k in range(n_starts)]
        if device_cem and cem_multi is not None:
            stage1 = cem_multi(x0, sigma0, seeds, **cem)     # every run on the device, one host call (NativeEvaluator.cem_multi)
        else:
            stage1 = _cem_optimize_multi(objective_batch, x0, sigma0, seeds, **cem)
        stage2 = _refine_with_gradient(value_and_grad, np.array([r["x"] for r in stage1]), sigma0, seed=seed, **grad_opt)
        lead = min(stage1, key=lambda r: r["fun"])
        nit1 = max(int(r["nit"]) for r in stage1)
        if stage2["fun"] <= float(lead["fun"]):
            return {"x": stage2["x"], "success": True, "message": "Hybrid CEM+gradient: refined", "fun": stage2["fun"],
                    "nit": nit1 + int(stage2["nit"])}
        return {"x": lead["x"], "success": True, "message": "Hybrid CEM+gradient: kept CEM result", "nit": nit1,
                "fun": float(lead["fun"])}

    if name in {"CEM-POWELL", "HYBRID"}:
        cem, powell = dict(_HYBRID_CEM_DEFAULTS), dict(_HYBRID_POWELL_DEFAULTS)
        if options:
            if isinstance(options.get("cem"), dict):
                cem.update(options["cem"])
            if isinstance(options.get("powell"), dict):
                powell.update(options["powell"])
            for key, val in options.items():
                if key in cem:
                    cem[key] = val
                if key in powell:
                    powell[key] = val
        stage1 = _cem_optimize(objective_batch, x0=x0, sigma0=sigma0, **cem)
        stage2 = minimize(scalar, stage1["x"], method="Powell", options=powell)
        if float(stage2.fun) <= float(stage1["fun"]):
            return {"x": stage2.x, "success": bool(stage2.success), "message": f"Hybrid CEM+Powell: {stage2.message}",
                    "nit": int(getattr(stage2, "nit", 0)) + int(stage1["nit"]), "fun": float(stage2.fun)}
        return {"x": stage1["x"], "success": bool(stage1["success"]), "message": "Hybrid CEM+Powell: kept CEM result",
                "nit": int(stage1["nit"]), "fun": float(stage1["fun"])}

    if name in {"NELDER", "NEAD-MELDER", "NEAD_MELDER"}:
        method, name = "Nelder-Mead", "NELDER-MEAD"
    if name == "POWELL":
        scipy_options = {"maxiter": 220, "xtol": 1e-3, "ftol": 1e-3}
    elif name == "NELDER-MEAD":
        scipy_options = {"maxiter": 300, "xatol": 1e-3, "fatol": 1e-3}
    else:
        scipy_options = {"maxiter": 250, "xatol": 1e-3, "fatol": 1e-3}
    if options:
        nested = options.get(method.lower().replace("-", "_"))
        scipy_options.update(nested if isinstance(nested, dict) else options)
    res = minimize(scalar, x0, method=method, options=scipy_options)
    return {"x": res.x, "success": bool(res.success), "message": str(res.message), "nit": int(getattr(res, "nit", 0)),
            "fun": float(res.fun)}


def _make_evaluator(problem: PathProblem, evaluator_factory: Optional[EvaluatorFactory], device: int, dtype):
    if evaluator_factory is not None:
        return evaluator_factory(problem)
    return NativeEvaluator(problem, device=device, dtype=dtype)   # raises without the HIP library / a GPU


# ---------------------------------------------------------------------------------------------
# planner entry points
# ---------------------------------------------------------------------------------------------


def optimize_bspline_path(
    scene,
    start: np.ndarray,
    goal: np.ndarray,
    n_vias: int = 3,
    tool_half_extents: Tuple[float, float, float] = (0.0, 0.0, 0.0),
    moving_block_size: Optional[Tuple[float, float, float]] = None,
    moving_block_quat: Tuple[float, float, float, float] = (0.0, 0.0, 0.0, 1.0),
    collision_ignore_ids: Optional[List[str]] = None,
    safety_margin: float = 0.01,
    n_samples_curve: int = 121,
    collision_check_subsample: int = 1,
    start_yaw_deg: float = 0.0,
    goal_yaw_deg: float = 0.0,
    n_yaw_vias: int = 0,
    combined_4d: bool = True,
    w_len: float = 1.0,
    w_curv: float = 0.1,
    w_yaw_smooth: float = 0.0,
    w_safe: float = 50.0,
    preferred_safety_margin: Optional[float] = None,
    relax_preferred_final_fraction: float = 0.0,
    w_safe_preferred: float = 0.0,
    w_approach_rebound: float = 0.0,
    w_goal_clearance: float = 0.0,
    goal_clearance_target: Optional[float] = None,
    w_goal_clearance_target: float = 0.0,
    approach_only_clearance: Optional[float] = None,
    contact_window_fraction: float = 0.1,
    w_approach_clearance: float = 0.0,
    w_approach_collision: float = 0.0,
    approach_fraction: float = 0.2,
    w_via_dev: float = 0.0,
    w_yaw_dev: float = 0.0,
    w_yaw_monotonic: float = 0.0,
    yaw_goal_reach_u: float = 1.0,
    w_yaw_schedule: float = 0.0,
    goal_approach_normals: Optional[np.ndarray] = None,
    goal_approach_window_fraction: float = 0.1,
    w_goal_approach_normal: float = 0.0,
    init_offset_scale: float = 1.0,
    method: str = "Powell",
    options: Optional[Dict] = None,
    *,
    device: int = 0,
    dtype=np.float64,
    evaluator_factory: Optional[EvaluatorFactory] = None,
) -> Tuple[Callable[[np.ndarray], np.ndarray], np.ndarray, Dict]:
    """Optimise all vias (and yaw controls) of the cubic B-spline through [start, vias..., goal].

    Same contract as the reference's function of this name.  ``moving_block_quat`` is accepted and
    unused, as in the reference (per-sample yaw quaternions always take precedence, :538-545).
    Keyword-only extras: ``device`` (GPU index), ``dtype`` (float64 / float32 kernels) and
    ``evaluator_factory`` (dependency injection for tests; the default is the HIP library)."""
    from . import _native

    problem = build_path_problem(
        scene, start, goal, _native.build_basis, n_vias=n_vias, tool_half_extents=tool_half_extents,
        moving_block_size=moving_block_size, collision_ignore_ids=collision_ignore_ids, safety_margin=safety_margin,
        n_samples_curve=n_samples_curve, collision_check_subsample=collision_check_subsample,
        start_yaw_deg=start_yaw_deg, goal_yaw_deg=goal_yaw_deg, n_yaw_vias=n_yaw_vias, combined_4d=combined_4d,
        w_len=w_len, w_curv=w_curv, w_yaw_smooth=w_yaw_smooth, w_safe=w_safe,
        preferred_safety_margin=preferred_safety_margin, relax_preferred_final_fraction=relax_preferred_final_fraction,
        w_safe_preferred=w_safe_preferred, w_approach_rebound=w_approach_rebound, w_goal_clearance=w_goal_clearance,
        goal_clearance_target=goal_clearance_target, w_goal_clearance_target=w_goal_clearance_target,
        approach_only_clearance=approach_only_clearance, contact_window_fraction=contact_window_fraction,
        w_approach_clearance=w_approach_clearance, w_approach_collision=w_approach_collision,
        approach_fraction=approach_fraction, w_via_dev=w_via_dev, w_yaw_dev=w_yaw_dev, w_yaw_monotonic=w_yaw_monotonic,
        yaw_goal_reach_u=yaw_goal_reach_u, w_yaw_schedule=w_yaw_schedule, goal_approach_normals=goal_approach_normals,
        goal_approach_window_fraction=goal_approach_window_fraction, w_goal_approach_normal=w_goal_approach_normal,
        init_offset_scale=init_offset_scale,
    )
    start = np.asarray(start, float).reshape(3)
    goal = np.asarray(goal, float).reshape(3)
    evaluate = _make_evaluator(problem, evaluator_factory, device, dtype)
    try:
        return _solve_path(problem, evaluate, start, goal, start_yaw_deg, goal_yaw_deg, combined_4d, method, options,
                           goal_clearance_target, approach_only_clearance, contact_window_fraction,
                           goal_approach_window_fraction, goal_approach_normals, yaw_goal_reach_u)
    finally:
        _close(evaluate)   # a failed solve (SciPy, NativeError) must not keep a GPU context and its buffers alive


def _solve_path(problem, evaluate, start, goal, start_yaw_deg, goal_yaw_deg, combined_4d, method, options,
                goal_clearance_target, approach_only_clearance, contact_window_fraction, goal_approach_window_fraction,
                goal_approach_normals, yaw_goal_reach_u):
    """Solve + post-solve bookkeeping of optimize_bspline_path (ref :643-746) on an open evaluator."""
    n_pos = 3 * problem.n_free

    def objective_single(x):
        return (float(evaluate(np.asarray(x, dtype=float)[None, :])["cost"][0]),)

    def objective_batch(X):
        return np.asarray(evaluate(X)["cost"], dtype=float)

    def value_and_grad(X):
        out = evaluate(X, want_grad=True)
        return np.asarray(out["cost"], dtype=float), np.asarray(out["grad"], dtype=float)

    opt = _solve_optimizer(objective_single, x0=problem.x0, sigma0=problem.sigma0, method=method, options=options,
                           objective_batch=objective_batch, value_and_grad=value_and_grad,
                           cem_multi=getattr(evaluate, "cem_multi", None))
    x_opt = np.asarray(opt["x"], dtype=float)
    vias_opt = x_opt[:n_pos].reshape(problem.n_free, 3)
    if problem.n_yaw_free > 0:
        yaw_ctrl_opt = np.hstack([[float(start_yaw_deg)], x_opt[n_pos:], [float(goal_yaw_deg)]])
    else:
        yaw_ctrl_opt = np.array([float(start_yaw_deg), float(goal_yaw_deg)], dtype=float)

    # callables over arbitrary u (SciPy on the host, as the reference does after the solve, :648-675)
    us = np.linspace(0.0, 1.0, problem.n_samples)
    if combined_4d:
        S4 = build_cubic_bspline(np.hstack([np.vstack([start, vias_opt, goal]), yaw_ctrl_opt.reshape(-1, 1)]))

        def S_opt(uq):
            q = np.asarray(S4(uq), dtype=float)
            return q[:3].reshape(1, 3) if q.ndim == 1 else q[:, :3]

        def S_yaw_opt(uq):
            q = np.asarray(S4(uq), dtype=float)
            return np.array([q[3]], dtype=float) if q.ndim == 1 else q[:, 3]

        Q = S4(us)
        P_opt, yaw_samples_opt = Q[:, :3], Q[:, 3]
    else:
        S_opt = build_cubic_bspline(np.vstack([start, vias_opt, goal]))
        S_yaw_opt = build_scalar_bspline(yaw_ctrl_opt)
        P_opt, yaw_samples_opt = S_opt(us), S_yaw_opt(us)

    final = evaluate(x_opt[None, :], want_terms=True)
    terms = np.asarray(final["terms"], dtype=float)[0]
    d_opt = evaluate.clearance_profile(np.ascontiguousarray(P_opt), np.ascontiguousarray(yaw_samples_opt))

    info = {"success": opt["success"], "message": opt["message"], "fun": opt["fun"]}
    for name, value in zip(abi.TERM_NAMES, terms):
        info[name] = float(value)
    info.update({
        "min_clearance": float(np.min(d_opt)),
        "mean_clearance": float(np.mean(d_opt)),
        "turn_angle_mean_deg": mean_turn_angle_deg(P_opt),
        "yaw_start_deg": float(start_yaw_deg),
        "yaw_goal_deg": float(goal_yaw_deg),
        "yaw_ctrl_deg": yaw_ctrl_opt.copy(),
        "yaw_samples_deg": np.array(yaw_samples_opt, dtype=float, copy=True),
        "yaw_fn": S_yaw_opt,
        "combined_4d": bool(combined_4d),
        "solver_method": method,
        "required_clearance": problem.required_clearance,
        "preferred_clearance": problem.preferred_clearance,
        "goal_clearance_target": goal_clearance_target,
        "approach_only_clearance": approach_only_clearance,
        "contact_window_fraction": float(contact_window_fraction),
        "goal_approach_window_fraction": float(goal_approach_window_fraction),
        "goal_approach_normals": None if goal_approach_normals is None else np.asarray(goal_approach_normals, dtype=float).copy(),
        "yaw_goal_reach_u": float(yaw_goal_reach_u),
        "collision_model": "box" if problem.model == abi.MP_MODEL_BOX else "point",
        "nit": opt["nit"],
    })
    return S_opt, vias_opt, info


def _close(evaluate) -> None:
    fn = getattr(evaluate, "close", None)
    if callable(fn):
        fn()


def optimize_bspline_with_vias(
    scene,
    start: np.ndarray,
    via: np.ndarray,
    goal: np.ndarray,
    n_additional_vias: int = 2,
    tool_half_extents: Tuple[float, float, float] = (0.0, 0.0, 0.0),
    safety_margin: float = 0.01,
    n_samples_curve: int = 121,
    collision_check_subsample: int = 1,
    w_len: float = 1.0,
    w_curv: float = 0.1,
    w_safe: float = 50.0,
    init_offset_scale: float = 1.0,
    method: str = "CEM",
    options: Optional[Dict] = None,
    *,
    device: int = 0,
    dtype=np.float64,
    evaluator_factory: Optional[EvaluatorFactory] = None,
) -> Tuple[Callable[[np.ndarray], np.ndarray], np.ndarray, Dict]:
    """Optimise N additional vias of the spline through [start, via, vias..., goal] for length,
    curvature and the safety margin (3-term objective; same contract as the reference, :749-849)."""
    from . import _native

    problem = build_vias_problem(
        scene, start, via, goal, _native.build_basis, n_additional_vias=n_additional_vias,
        tool_half_extents=tool_half_extents, safety_margin=safety_margin, n_samples_curve=n_samples_curve,
        collision_check_subsample=collision_check_subsample, w_len=w_len, w_curv=w_curv, w_safe=w_safe,
        init_offset_scale=init_offset_scale,
    )
    start = np.asarray(start, float).reshape(3)
    via = np.asarray(via, float).reshape(3)
    goal = np.asarray(goal, float).reshape(3)
    evaluate = _make_evaluator(problem, evaluator_factory, device, dtype)
    try:
        return _solve_vias(problem, evaluate, start, via, goal, n_additional_vias, n_samples_curve, method, options)
    finally:
        _close(evaluate)


def _solve_vias(problem, evaluate, start, via, goal, n_additional_vias, n_samples_curve, method, options):
    """Solve + post-solve bookkeeping of optimize_bspline_with_vias (ref :820-849) on an open evaluator."""

    def objective_single(x):
        return (float(evaluate(np.asarray(x, dtype=float)[None, :])["cost"][0]),)

    def objective_batch(X):
        return np.asarray(evaluate(X)["cost"], dtype=float)

    def value_and_grad(X):
        out = evaluate(X, want_grad=True)
        return np.asarray(out["cost"], dtype=float), np.asarray(out["grad"], dtype=float)

    opt = _solve_optimizer(objective_single, x0=problem.x0, sigma0=problem.sigma0, method=method, options=options,
                           objective_batch=objective_batch, value_and_grad=value_and_grad)
    x_opt = np.asarray(opt["x"], dtype=float)
    vias_opt = x_opt.reshape(n_additional_vias, 3)
    S_opt = build_cubic_bspline(np.vstack([start, via, vias_opt, goal]))
    P_opt, _ = sample_curve(S_opt, n=n_samples_curve)
    terms = np.asarray(evaluate(x_opt[None, :], want_terms=True)["terms"], dtype=float)[0]
    d_opt = evaluate.clearance_profile(np.ascontiguousarray(P_opt), None)
    info = {
        "success": opt["success"],
        "message": opt["message"],
        "fun": opt["fun"],
        "length": float(terms[abi.T_LEN]),
        "curvature_cost": float(terms[abi.T_CURV]),
        "safety_cost": float(terms[abi.T_SAFE]),
        "min_clearance": float(np.min(d_opt)),
        "mean_clearance": float(np.mean(d_opt)),
        "turn_angle_mean_deg": mean_turn_angle_deg(P_opt),
        "required_clearance": problem.required_clearance,
        "collision_model": "box" if problem.model == abi.MP_MODEL_BOX else "point",
        "nit": opt["nit"],
    }
    return S_opt, vias_opt, info


def optimize_bspline_two_vias(
    scene,
    start: np.ndarray,
    via: np.ndarray,
    goal: np.ndarray,
    tool_half_extents: Tuple[float, float, float] = (0.0, 0.0, 0.0),
    safety_margin: float = 0.01,
    n_samples_curve: int = 121,
    w_len: float = 1.0,
    w_curv: float = 0.1,
    w_safe: float = 50.0,
    init_offset_scale: float = 0.3,
    method: str = "Nelder-Mead",
    options: Dict = None,
    **native_kwargs,
) -> Tuple[Callable[[np.ndarray], np.ndarray], np.ndarray, np.ndarray, Dict]:
    """Two-additional-vias wrapper kept for callers of the older API (ref :852-884)."""
    S, vias, info = optimize_bspline_with_vias(
        scene=scene, start=start, via=via, goal=goal, n_additional_vias=2, tool_half_extents=tool_half_extents,
        safety_margin=safety_margin, n_samples_curve=n_samples_curve, w_len=w_len, w_curv=w_curv, w_safe=w_safe,
        init_offset_scale=init_offset_scale, method=method, options=options, **native_kwargs,
    )
    return S, vias[0], vias[1], info
