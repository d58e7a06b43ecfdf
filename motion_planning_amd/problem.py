"""Host-side construction of the closure constants of the reference objective.

The reference builds its objective as a Python closure inside
``optimize_bspline_path`` (geom/spline_opt.py:438-499) and
``optimize_bspline_with_vias`` (geom/spline_opt.py:776-796).  Here the same constants
are derived once, with the reference's own NumPy expressions for every window and
mask (windows are integers derived on the host), and packed into
the POD descriptors of include/mp_api.h (``mp_config`` + ``mp_task``).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _abi as abi


def default_via_initialization(start: np.ndarray, goal: np.ndarray, n_vias: int) -> np.ndarray:
    """Vias spread uniformly over the start->goal segment (geom/spline_opt.py:214-219)."""
    if n_vias <= 0:
        return np.empty((0, 3), dtype=float)
    frac = np.linspace(1.0 / (n_vias + 1), n_vias / (n_vias + 1), n_vias)
    return start[None, :] + frac[:, None] * (goal - start)[None, :]


def _unit(v, eps: float = 1e-12) -> np.ndarray:
    """``_normalize`` of the reference (geom/spline_opt.py:109-114)."""
    a = np.asarray(v, dtype=float).reshape(-1)
    n = float(np.linalg.norm(a))
    if n < eps:
        return np.zeros_like(a)
    return a / n


def goal_normal_resultant(goal_normals) -> Tuple[np.ndarray, bool]:
    """Closure-constant half of ``_goal_approach_alignment_cost`` (spline_opt.py:134-142)."""
    if goal_normals is None:
        return np.zeros(3), False
    arr = np.asarray(goal_normals, dtype=float)
    if arr.size == 0:
        return np.zeros(3), False
    rows = arr.reshape(-1, 3)
    units = np.array([_unit(r) for r in rows], dtype=float)
    s = _unit(np.sum(units, axis=0))
    if not np.any(s):
        s = _unit(units[0])
    if not np.any(s):
        return np.zeros(3), False
    return s, True


def as_boxes10(boxes) -> np.ndarray:
    """Any accepted row format -> contiguous float64 (B,10) rows cx,cy,cz,hx,hy,hz,qx,qy,qz,qw (quaternion of the
    box's rotation, identity = 0,0,0,1): (B,6) axis aligned; (B,8) with the (z, w) of a yaw-only quaternion
    appended; (B,10) with the full unit quaternion."""
    b = np.asarray(boxes, dtype=np.float64)
    if b.size == 0:
        return np.zeros((0, 10), dtype=np.float64)
    b = np.atleast_2d(b)
    n = b.shape[0]
    if b.shape[1] == 6:
        b = np.hstack([b, np.zeros((n, 3)), np.ones((n, 1))])
    elif b.shape[1] == 8:
        b = np.hstack([b[:, :6], np.zeros((n, 2)), b[:, 6:8]])
    elif b.shape[1] != 10:
        raise ValueError("box rows must have 6, 8 or 10 columns")
    return np.ascontiguousarray(b)


def as_boxes8(boxes) -> np.ndarray:
    """(B,6) or (B,8) rows -> (B,8) rows cx,cy,cz,hx,hy,hz,qz,qw (yaw-only scenes; kept for callers of the 8-column form)."""
    b = as_boxes10(boxes)
    if np.any(b[:, 6] != 0.0) or np.any(b[:, 7] != 0.0):
        raise ValueError("tilted boxes need the 10-column form")
    return np.ascontiguousarray(np.hstack([b[:, :6], b[:, 8:10]]))


def scene_to_boxes(scene, ignore_ids: Optional[Sequence[str]] = None) -> np.ndarray:
    """SoA export of a duck-typed scene: rows (cx,cy,cz,hx,hy,hz) of every block that is
    not ignored (geom/scene.py:88-95).  If any static block is rotated about z, the rows get two
    more columns, the z and w of its quaternion (geom/blocks.py:8-21 keeps ``quat`` as [x,y,z,w]); if any
    block is tilted out of the horizontal plane, four more: the whole quaternion."""
    ignore = set(ignore_ids or [])
    rows: List[List[float]] = []
    rotated = tilted = False
    for b in getattr(scene, "blocks", []):
        oid = getattr(b, "object_id", None)
        if oid is not None and oid in ignore:
            continue
        q = tuple(float(v) for v in getattr(b, "quat", (0.0, 0.0, 0.0, 1.0)))
        if abs(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] - 1.0) > 1e-9:
            raise ValueError("static block quaternion is not of unit length")
        if q[0] != 0.0 or q[1] != 0.0:
            tilted = True
        elif q[2] != 0.0:
            rotated = True
        else:
            q = (0.0, 0.0, 0.0, 1.0)   # w = -1 is the identity too
        sx, sy, sz = (float(v) for v in b.size)
        px, py, pz = (float(v) for v in b.position)
        rows.append([px, py, pz, 0.5 * sx, 0.5 * sy, 0.5 * sz, q[0], q[1], q[2], q[3]])
    if not rows:
        return np.zeros((0, 6), dtype=np.float64)
    arr = np.ascontiguousarray(np.asarray(rows, dtype=np.float64))
    if tilted:
        return arr
    if rotated:
        return np.ascontiguousarray(np.hstack([arr[:, :6], arr[:, 8:10]]))
    return np.ascontiguousarray(arr[:, :6])


@dataclass
class PathProblem:
    """All closure constants of one objective, ready for the C ABI."""

    n_samples: int
    n_ctrl: int
    first_free: int
    n_free: int
    n_yaw_ctrl: int
    n_yaw_free: int
    model: int
    idx_safe: np.ndarray
    last_if_not_close: int
    keep_n: int
    use_pref: int
    tail_start: int
    use_goal_target: int
    goal_tail_n: int
    term_mask: int
    weights: np.ndarray
    required_clearance: float
    preferred_clearance: float
    approach_target: float
    goal_clearance_target: float
    point_radius: float
    half_extents: np.ndarray
    approach_mask: np.ndarray
    basis: np.ndarray
    basis_yaw: np.ndarray
    t_sched: np.ndarray
    ctrl: np.ndarray            # (C,3): fixed rows + via_init in the free rows
    yaw_ctrl: np.ndarray        # (Cy,): yaw_ctrl_ref
    goal_normal: np.ndarray
    has_goal_normal: bool
    yaw_increasing: bool
    boxes: np.ndarray           # (B,6) axis aligned, (B,8) with (qz, qw) of each box's yaw quaternion, or (B,10) with (qx,qy,qz,qw)
    x0: np.ndarray
    sigma0: np.ndarray
    combined_4d: bool = True
    _keep: list = field(default_factory=list, repr=False)

    @property
    def dim(self) -> int:
        return 3 * self.n_free + self.n_yaw_free

    @property
    def n_safe(self) -> int:
        return int(self.idx_safe.size)

    def config_struct(self) -> abi.MpConfig:
        cfg = abi.MpConfig()
        cfg.n_samples = self.n_samples
        cfg.n_ctrl = self.n_ctrl
        cfg.first_free = self.first_free
        cfg.n_free = self.n_free
        cfg.n_yaw_ctrl = self.n_yaw_ctrl
        cfg.n_yaw_free = self.n_yaw_free
        cfg.model = self.model
        cfg.n_safe = self.n_safe
        cfg.last_if_not_close = self.last_if_not_close
        cfg.keep_n = self.keep_n
        cfg.use_pref = self.use_pref
        cfg.tail_start = self.tail_start
        cfg.use_goal_target = self.use_goal_target
        cfg.goal_tail_n = self.goal_tail_n
        cfg.term_mask = self.term_mask
        for i in range(abi.MP_NTERMS):
            cfg.weights[i] = float(self.weights[i])
        cfg.required_clearance = self.required_clearance
        cfg.preferred_clearance = self.preferred_clearance
        cfg.approach_target = self.approach_target
        cfg.goal_clearance_target = self.goal_clearance_target
        cfg.point_radius = self.point_radius
        for i in range(3):
            cfg.half_extents[i] = float(self.half_extents[i])
        idx = np.ascontiguousarray(self.idx_safe, dtype=np.int32)
        msk = np.ascontiguousarray(self.approach_mask, dtype=np.uint8)
        bas = np.ascontiguousarray(self.basis, dtype=np.float64)
        basy = np.ascontiguousarray(self.basis_yaw, dtype=np.float64)
        tsc = np.ascontiguousarray(self.t_sched, dtype=np.float64)
        self._keep = [idx, msk, bas, basy, tsc]
        cfg.idx_safe = idx.ctypes.data_as(C.POINTER(C.c_int32))
        cfg.approach_mask = msk.ctypes.data_as(C.POINTER(C.c_uint8))
        cfg.basis = bas.ctypes.data_as(C.POINTER(C.c_double))
        cfg.basis_yaw = basy.ctypes.data_as(C.POINTER(C.c_double))
        cfg.t_sched = tsc.ctypes.data_as(C.POINTER(C.c_double))
        return cfg

    def task_struct(self, box_begin: int = 0, box_count: Optional[int] = None) -> abi.MpTask:
        t = abi.MpTask()
        for j in range(self.n_ctrl):
            for a in range(3):
                t.ctrl[j][a] = float(self.ctrl[j, a])
        for j in range(self.n_yaw_ctrl):
            t.yaw_ctrl[j] = float(self.yaw_ctrl[j])
        for a in range(3):
            t.goal_normal[a] = float(self.goal_normal[a])
        t.has_goal_normal = int(self.has_goal_normal)
        t.yaw_increasing = int(self.yaw_increasing)
        t.box_begin = int(box_begin)
        t.box_count = int(self.boxes.shape[0] if box_count is None else box_count)
        return t


def _check_sizes(n_samples: int, n_ctrl: int, n_yaw_ctrl: int) -> None:
    if n_samples > abi.MP_MAX_SAMPLES:
        raise ValueError(f"n_samples_curve must be <= {abi.MP_MAX_SAMPLES}")
    if n_samples < 2:
        raise ValueError("n_samples_curve must be >= 2")
    if n_ctrl > abi.MP_MAX_CTRL or n_yaw_ctrl > abi.MP_MAX_CTRL:
        raise ValueError(f"at most {abi.MP_MAX_CTRL} waypoints are supported")


def build_path_problem(
    scene,
    start,
    goal,
    basis_fn: Callable[[int, int, int], np.ndarray],
    n_vias: int = 3,
    tool_half_extents=(0.0, 0.0, 0.0),
    moving_block_size=None,
    collision_ignore_ids=None,
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
    preferred_safety_margin=None,
    relax_preferred_final_fraction: float = 0.0,
    w_safe_preferred: float = 0.0,
    w_approach_rebound: float = 0.0,
    w_goal_clearance: float = 0.0,
    goal_clearance_target=None,
    w_goal_clearance_target: float = 0.0,
    approach_only_clearance=None,
    contact_window_fraction: float = 0.1,
    w_approach_clearance: float = 0.0,
    w_approach_collision: float = 0.0,
    approach_fraction: float = 0.2,
    w_via_dev: float = 0.0,
    w_yaw_dev: float = 0.0,
    w_yaw_monotonic: float = 0.0,
    yaw_goal_reach_u: float = 1.0,
    w_yaw_schedule: float = 0.0,
    goal_approach_normals=None,
    goal_approach_window_fraction: float = 0.1,
    w_goal_approach_normal: float = 0.0,
    init_offset_scale: float = 1.0,
) -> PathProblem:
    """Closure constants of ``optimize_bspline_path`` (geom/spline_opt.py:438-499 and the
    window rules inside objective_single, :523-603).  Raises the reference's ValueErrors."""
    start = np.asarray(start, float).reshape(3)
    goal = np.asarray(goal, float).reshape(3)
    if n_vias < 2:
        raise ValueError("n_vias must be >= 2 (cubic spline needs >=4 points total).")
    if collision_check_subsample < 1:
        raise ValueError("collision_check_subsample must be >= 1")
    if n_yaw_vias < 0:
        raise ValueError("n_yaw_vias must be >= 0")
    if combined_4d and n_yaw_vias not in (0, n_vias):
        raise ValueError("For combined_4d=True, n_yaw_vias must be 0 or equal to n_vias.")
    if not (0.0 < float(approach_fraction) <= 1.0):
        raise ValueError("approach_fraction must be in (0, 1].")
    if not (0.0 < float(contact_window_fraction) < 1.0):
        raise ValueError("contact_window_fraction must be in (0, 1).")
    if not (0.0 < float(yaw_goal_reach_u) <= 1.0):
        raise ValueError("yaw_goal_reach_u must be in (0, 1].")
    if not (0.0 <= float(relax_preferred_final_fraction) < 1.0):
        raise ValueError("relax_preferred_final_fraction must be in [0, 1).")
    if not (0.0 < float(goal_approach_window_fraction) <= 1.0):
        raise ValueError("goal_approach_window_fraction must be in (0, 1].")

    if moving_block_size is None and any(float(v) > 0.0 for v in tool_half_extents):
        moving_block_size = tuple(2.0 * float(v) for v in tool_half_extents)

    S = int(n_samples_curve)
    n_ctrl = int(n_vias) + 2
    yaw_free = int(n_vias) if combined_4d else int(n_yaw_vias)
    n_yaw_ctrl = yaw_free + 2
    _check_sizes(S, n_ctrl, n_yaw_ctrl)

    req = float(safety_margin)
    pref = float(preferred_safety_margin) if preferred_safety_margin is not None else req
    pref = max(pref, req)

    via_init = default_via_initialization(start, goal, int(n_vias))
    x0_pos = via_init.reshape(-1)
    yaw_ref = np.linspace(float(start_yaw_deg), float(goal_yaw_deg), n_yaw_ctrl, dtype=float)
    x0 = np.hstack([x0_pos, yaw_ref[1:-1]]) if yaw_free > 0 else x0_pos.copy()
    sigma_base = np.linalg.norm(goal - start) * init_offset_scale / max(n_vias, 1)
    sigma0 = np.full_like(x0_pos, max(0.05, sigma_base), dtype=float)
    if yaw_free > 0:
        sigma0 = np.hstack([sigma0, np.full((yaw_free,), 20.0, dtype=float)])

    us = np.linspace(0.0, 1.0, S)
    # collision samples (:523-533)
    if collision_check_subsample > 1:
        idx = np.arange(0, S, int(collision_check_subsample), dtype=int)
        if idx[-1] != S - 1:
            idx = np.append(idx, S - 1)
    else:
        idx = np.arange(S, dtype=int)
    n_safe = int(idx.size)
    us_safe = us[idx]

    use_pref = bool(pref > req and w_safe_preferred > 0.0)
    if relax_preferred_final_fraction > 0.0:
        keep_n = max(1, int(np.floor((1.0 - relax_preferred_final_fraction) * n_safe)))
    else:
        keep_n = n_safe
    n_tail = max(3, int(np.ceil(float(approach_fraction) * n_safe)))
    tail_start = max(0, n_safe - n_tail)
    mask = us_safe < (1.0 - float(contact_window_fraction))
    if not np.any(mask):
        mask = np.ones(n_safe, dtype=bool)
        if n_safe > 1:
            mask[-1] = False
    approach_target = pref if approach_only_clearance is None else float(approach_only_clearance)
    use_goal_target = bool(goal_clearance_target is not None and w_goal_clearance_target > 0.0)
    goal_tail_n = max(3, int(np.ceil(float(goal_approach_window_fraction) * S)))
    t_sched = np.clip(us / float(yaw_goal_reach_u), 0.0, 1.0)

    basis = np.ascontiguousarray(basis_fn(n_ctrl, 3, S), dtype=np.float64)
    if combined_4d:
        basis_yaw = basis
    else:
        basis_yaw = np.ascontiguousarray(basis_fn(n_yaw_ctrl, min(3, n_yaw_ctrl - 1), S), dtype=np.float64)

    s_res, has_normal = goal_normal_resultant(goal_approach_normals)
    weights = np.array(
        [w_len, w_curv, w_yaw_smooth, w_safe, w_safe_preferred, w_approach_rebound, w_goal_clearance,
         w_goal_clearance_target, w_approach_clearance, w_approach_collision, w_via_dev, w_yaw_dev,
         w_yaw_monotonic, w_yaw_schedule, w_goal_approach_normal],
        dtype=float,
    )
    box_model = moving_block_size is not None
    half = 0.5 * np.asarray(moving_block_size if box_model else (0.0, 0.0, 0.0), dtype=float)
    # Scene.signed_distance has no ignore list (geom/scene.py:50); the box query does.
    boxes = scene_to_boxes(scene, collision_ignore_ids if box_model else None)

    return PathProblem(
        n_samples=S, n_ctrl=n_ctrl, first_free=1, n_free=int(n_vias), n_yaw_ctrl=n_yaw_ctrl,
        n_yaw_free=yaw_free, model=abi.MP_MODEL_BOX if box_model else abi.MP_MODEL_POINT,
        idx_safe=idx.astype(np.int32), last_if_not_close=0, keep_n=int(keep_n), use_pref=int(use_pref),
        tail_start=int(tail_start), use_goal_target=int(use_goal_target), goal_tail_n=int(goal_tail_n),
        term_mask=abi.TERM_MASK_ALL, weights=weights, required_clearance=req, preferred_clearance=pref,
        approach_target=float(approach_target),
        goal_clearance_target=float(goal_clearance_target) if goal_clearance_target is not None else 0.0,
        point_radius=1e-6, half_extents=half, approach_mask=mask.astype(np.uint8), basis=basis,
        basis_yaw=basis_yaw, t_sched=t_sched, ctrl=np.vstack([start, via_init, goal]), yaw_ctrl=yaw_ref,
        goal_normal=s_res, has_goal_normal=has_normal,
        yaw_increasing=bool(float(goal_yaw_deg) >= float(start_yaw_deg)), boxes=boxes, x0=x0,
        sigma0=sigma0, combined_4d=bool(combined_4d),
    )


def build_vias_problem(
    scene,
    start,
    via,
    goal,
    basis_fn: Callable[[int, int, int], np.ndarray],
    n_additional_vias: int = 2,
    tool_half_extents=(0.0, 0.0, 0.0),
    safety_margin: float = 0.01,
    n_samples_curve: int = 121,
    collision_check_subsample: int = 1,
    w_len: float = 1.0,
    w_curv: float = 0.1,
    w_safe: float = 50.0,
    init_offset_scale: float = 1.0,
) -> PathProblem:
    """Closure constants of ``optimize_bspline_with_vias`` (geom/spline_opt.py:776-819)."""
    start = np.asarray(start, float).reshape(3)
    via = np.asarray(via, float).reshape(3)
    goal = np.asarray(goal, float).reshape(3)
    if n_additional_vias < 1:
        raise ValueError("n_additional_vias must be >= 1")
    if collision_check_subsample < 1:
        raise ValueError("collision_check_subsample must be >= 1")
    size = None
    if any(float(v) > 0.0 for v in tool_half_extents):
        size = tuple(2.0 * float(v) for v in tool_half_extents)
    S = int(n_samples_curve)
    n_ctrl = 3 + int(n_additional_vias)
    _check_sizes(S, n_ctrl, 2)
    via_init = default_via_initialization(via, goal, int(n_additional_vias))
    x0 = via_init.reshape(-1).copy()
    sigma_base = np.linalg.norm(goal - via) * init_offset_scale / max(n_additional_vias, 1)
    sigma0 = np.full_like(x0, max(0.05, sigma_base), dtype=float)

    us = np.linspace(0.0, 1.0, S)
    last_cond = 0
    if collision_check_subsample > 1:
        idx = np.arange(0, S, int(collision_check_subsample), dtype=int)   # P[::k]
        if idx[-1] != S - 1:
            # appended unless allclose(P_safe[-1], P[-1]) — decided per candidate (:804-805)
            idx = np.append(idx, S - 1)
            last_cond = 1
    else:
        idx = np.arange(S, dtype=int)
    n_safe = int(idx.size)
    weights = np.zeros(abi.MP_NTERMS, dtype=float)
    weights[abi.T_LEN] = w_len
    weights[abi.T_CURV] = w_curv
    weights[abi.T_SAFE] = w_safe
    box_model = size is not None
    half = 0.5 * np.asarray(size if box_model else (0.0, 0.0, 0.0), dtype=float)
    basis = np.ascontiguousarray(basis_fn(n_ctrl, 3, S), dtype=np.float64)
    basis_yaw = np.ascontiguousarray(basis_fn(2, 1, S), dtype=np.float64)
    return PathProblem(
        n_samples=S, n_ctrl=n_ctrl, first_free=2, n_free=int(n_additional_vias), n_yaw_ctrl=2,
        n_yaw_free=0, model=abi.MP_MODEL_BOX if box_model else abi.MP_MODEL_POINT,
        idx_safe=idx.astype(np.int32), last_if_not_close=last_cond, keep_n=n_safe, use_pref=0,
        tail_start=max(0, n_safe - 3), use_goal_target=0, goal_tail_n=3, term_mask=abi.TERM_MASK_3TERM,
        weights=weights, required_clearance=float(safety_margin), preferred_clearance=float(safety_margin),
        approach_target=float(safety_margin), goal_clearance_target=0.0, point_radius=1e-6,
        half_extents=half, approach_mask=np.ones(n_safe, dtype=np.uint8), basis=basis, basis_yaw=basis_yaw,
        t_sched=np.clip(us, 0.0, 1.0), ctrl=np.vstack([start, via, via_init, goal]),
        yaw_ctrl=np.zeros(2), goal_normal=np.zeros(3), has_goal_normal=False, yaw_increasing=True,
        boxes=scene_to_boxes(scene, None), x0=x0, sigma0=sigma0, combined_4d=False,
    )
