"""ctypes binding of libmp_hip.so (include/mp_api.h).

There is deliberately no fallback: if the library is missing, or no MI355X/HIP device is
visible, the first call raises.  Build the library with ``__graft_entry__.build()`` (or
``make -C motion_planning_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from . import _abi as abi
from .problem import PathProblem, as_boxes10

import os

# MP_HIP_LIB lets kernel-tuning runs A/B another build of the same library
LIB_PATH = Path(os.environ.get("MP_HIP_LIB") or (Path(__file__).resolve().parent / "libmp_hip.so"))
_lib = None


class NativeError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"libmp_hip error {code}: {text}")
        self.code = code


def load():
    """Load libmp_hip.so and declare prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    # torch wheels bundle their own libamdhip64.so.7 / libhsa-runtime64: two HIP runtimes in
    # one process cannot both own the GPU.  Importing torch first makes the loader resolve
    # our NEEDED libamdhip64.so.7 to the copy torch already mapped, so both share one runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(LIB_PATH))
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.mp_api_version.restype = C.c_int
    lib.mp_last_error.restype = C.c_char_p
    lib.mp_last_error.argtypes = [vp]
    lib.mp_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    lib.mp_destroy.argtypes = [vp]
    lib.mp_set_stream.argtypes = [vp, vp]
    lib.mp_use_own_stream.argtypes = [vp]
    lib.mp_synchronize.argtypes = [vp]
    lib.mp_timer_start.argtypes = [vp]
    lib.mp_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    lib.mp_set_scene.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.mp_set_problem.argtypes = [vp, C.POINTER(abi.MpConfig), i32, C.POINTER(abi.MpTask)]
    lib.mp_build_basis.argtypes = [i32, i32, i32, vp]
    lib.mp_get_basis.argtypes = [vp, vp, vp]
    lib.mp_set_cands_per_task.argtypes = [vp, i64]
    lib.mp_eval.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    lib.mp_eval_noalloc.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    lib.mp_point_sdf.argtypes = [vp, i64, vp, C.c_double, i32, i32, vp, C.c_int]
    lib.mp_box_sdf.argtypes = [vp, i64, vp, vp, vp, i32, i32, vp, C.c_int]
    lib.mp_allgather_results.argtypes = [vp, vp, vp, vp, C.c_size_t]
    lib.mp_cem_solve.argtypes = [vp, i64, vp, vp, vp, C.POINTER(abi.MpCemOptions), vp, vp, vp, vp, vp, vp, C.c_int]
    for name in ("mp_create", "mp_destroy", "mp_set_stream", "mp_use_own_stream", "mp_synchronize", "mp_timer_start", "mp_timer_stop",
                 "mp_set_scene", "mp_set_problem", "mp_build_basis", "mp_get_basis", "mp_set_cands_per_task",
                 "mp_eval", "mp_eval_noalloc", "mp_point_sdf", "mp_box_sdf", "mp_cem_solve", "mp_allgather_results"):
        getattr(lib, name).restype = C.c_int
    if lib.mp_api_version() != abi.MP_API_VERSION:
        raise ImportError("libmp_hip.so was built from a different mp_api.h version")
    _lib = lib
    return lib


def build_basis(n_ctrl: int, degree: int, n_samples: int) -> np.ndarray:
    """Sampling matrix M (S x n) of the interpolating spline: host-side, no GPU needed."""
    lib = load()
    M = np.empty((n_samples, n_ctrl), dtype=np.float64)
    rc = lib.mp_build_basis(n_ctrl, degree, n_samples, M.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise NativeError(rc, f"mp_build_basis({n_ctrl}, {degree}, {n_samples}) rejected its arguments")
    return M


_SMALL_CALL = 2048   # rows up to which Context.eval reuses its own output buffers (and their ctypes pointers)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context:
    """One mp_ctx: a GPU, a dtype, a stream, a scene and a problem."""

    def __init__(self, device: int = 0, dtype=np.float64):
        self.lib = load()
        self.np_dtype = np.dtype(dtype)
        if self.np_dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("dtype must be float64 or float32")
        self._h = C.c_void_p()
        code = abi.MP_F64 if self.np_dtype == np.float64 else abi.MP_F32
        rc = self.lib.mp_create(int(device), code, C.byref(self._h))
        if rc != 0:
            raise NativeError(rc, (self.lib.mp_last_error(None) or b"").decode())
        self.device = int(device)
        self.problem: Optional[PathProblem] = None
        self.n_tasks = 0
        self.n_boxes = 0

    # -- plumbing --
    def _check(self, rc: int) -> None:
        if rc != 0:
            raise NativeError(rc, (self.lib.mp_last_error(self._h) or b"").decode())

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.mp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: Optional[int]) -> None:
        """An int is a hipStream_t handle used as given (0 = the default stream, torch's default on
        ROCm); None goes back to the context's own stream."""
        if stream_ptr is None:
            self._check(self.lib.mp_use_own_stream(self._h))
        else:
            self._check(self.lib.mp_set_stream(self._h, C.c_void_p(int(stream_ptr))))

    def synchronize(self) -> None:
        self._check(self.lib.mp_synchronize(self._h))

    def timer_start(self) -> None:
        self._check(self.lib.mp_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(self.lib.mp_timer_stop(self._h, C.byref(ms)))
        return float(ms.value)

    # -- scene / problem --
    def set_scene(self, boxes: np.ndarray) -> None:
        """(B,6) rows cx,cy,cz,hx,hy,hz; (B,8) with the (z, w) of each box's yaw-only quaternion appended; or (B,10)
        with the full quaternion (x, y, z, w) of each box."""
        b = as_boxes10(boxes)
        cols = [np.ascontiguousarray(b[:, i]) for i in range(6)]
        quat = None
        if b.shape[0] and not (np.all(b[:, 6:9] == 0.0) and np.all(b[:, 9] == 1.0)):
            quat = np.ascontiguousarray(b[:, 6:10])
        self._check(self.lib.mp_set_scene(self._h, b.shape[0], *[_p(c) for c in cols], _p(quat)))
        self.n_boxes = int(b.shape[0])

    def set_problem(self, problems: Sequence[PathProblem] | PathProblem, upload_scene: bool = True) -> None:
        """One problem, or several that share the config and differ in task data / boxes."""
        if isinstance(problems, PathProblem):
            problems = [problems]
        p0 = problems[0]
        if upload_scene:
            self.set_scene(np.vstack([as_boxes10(p.boxes) for p in problems]))
        cfg = p0.config_struct()
        tasks = (abi.MpTask * len(problems))()
        off = 0
        for i, p in enumerate(problems):
            tasks[i] = p.task_struct(box_begin=off, box_count=p.boxes.shape[0])
            off += p.boxes.shape[0]
        self._check(self.lib.mp_set_problem(self._h, C.byref(cfg), len(problems), tasks))
        self.problem = p0
        self.n_tasks = len(problems)
        self._small = None     # output buffers of the small-call path (eval), sized for this problem

    def set_cands_per_task(self, n: int) -> None:
        self._check(self.lib.mp_set_cands_per_task(self._h, int(n)))

    def get_basis(self):
        p = self.problem
        M = np.empty((p.n_samples, p.n_ctrl), dtype=np.float64)
        My = np.empty((p.n_samples, p.n_yaw_ctrl), dtype=np.float64)
        self._check(self.lib.mp_get_basis(self._h, _p(M), _p(My)))
        return M, My

    # -- hot path, host arrays --
    def eval(self, X: np.ndarray, task_of: Optional[np.ndarray] = None, want_terms: bool = True,
             want_grad: bool = False, want_dist: bool = False):
        p = self.problem
        if p is None:
            raise RuntimeError("set_problem first")
        dt = self.np_dtype
        X = np.ascontiguousarray(X, dtype=dt)
        if X.ndim == 1:
            X = X.reshape(1, -1)
        n, D = X.shape
        if D != p.dim:
            raise ValueError(f"X must have {p.dim} columns")
        if n <= _SMALL_CALL and task_of is None:
            # Small calls (one x from SciPy's Powell, one CEM population) are dominated by Python: ctypes pointer objects cost
            # ~3.5 us each and a call passes nine.  Outputs go to buffers allocated once per problem, whose pointers are kept;
            # the caller gets copies of the rows that were written.
            sm = getattr(self, "_small", None)
            if sm is None:
                cap = _SMALL_CALL
                arrs = {"cost": np.empty(cap, dtype=dt), "terms": np.empty((cap, abi.MP_NTERMS), dtype=dt),
                        "grad": np.empty((cap, D), dtype=dt), "min_clear": np.empty(cap, dtype=dt),
                        "flags": np.empty(cap, dtype=np.uint8), "dist": None}
                sm = self._small = {"arr": arrs, "ptr": {k: _p(v) for k, v in arrs.items()}}
            arr, ptr = sm["arr"], sm["ptr"]
            if want_dist and arr["dist"] is None:
                arr["dist"] = np.empty((_SMALL_CALL, p.n_safe), dtype=dt)
                ptr["dist"] = _p(arr["dist"])
            rc = self.lib.mp_eval(self._h, n, C.c_void_p(X.__array_interface__["data"][0]), None, ptr["cost"],
                                  ptr["terms"] if want_terms else None, ptr["grad"] if want_grad else None, ptr["min_clear"],
                                  ptr["flags"], ptr["dist"] if want_dist else None, abi.MP_MEM_HOST)
            if rc != 0:
                self._check(rc)
            return {"cost": arr["cost"][:n].copy(), "terms": arr["terms"][:n].copy() if want_terms else None,
                    "grad": arr["grad"][:n].copy() if want_grad else None, "min_clear": arr["min_clear"][:n].copy(),
                    "flags": arr["flags"][:n].copy(), "dist": arr["dist"][:n].copy() if want_dist else None}
        cost = np.empty(n, dtype=dt)
        terms = np.empty((n, abi.MP_NTERMS), dtype=dt) if want_terms else None
        grad = np.empty((n, D), dtype=dt) if want_grad else None
        minc = np.empty(n, dtype=dt)
        flags = np.empty(n, dtype=np.uint8)
        dist = np.empty((n, p.n_safe), dtype=dt) if want_dist else None
        tof = None if task_of is None else np.ascontiguousarray(task_of, dtype=np.int32)
        self._check(self.lib.mp_eval(self._h, n, _p(X), _p(tof), _p(cost), _p(terms), _p(grad), _p(minc), _p(flags),
                                     _p(dist), abi.MP_MEM_HOST))
        return {"cost": cost, "terms": terms, "grad": grad, "min_clear": minc, "flags": flags, "dist": dist}

    # -- hot path, device pointers (torch tensors' data_ptr()) --
    def eval_device(self, n: int, X_ptr: int, cost_ptr: int, terms_ptr: int = 0, grad_ptr: int = 0,
                    minc_ptr: int = 0, flags_ptr: int = 0, dist_ptr: int = 0, task_of_ptr: int = 0) -> None:
        vp = C.c_void_p
        self._check(self.lib.mp_eval(self._h, int(n), vp(X_ptr), vp(task_of_ptr or 0), vp(cost_ptr), vp(terms_ptr or 0),
                                     vp(grad_ptr or 0), vp(minc_ptr or 0), vp(flags_ptr or 0), vp(dist_ptr or 0),
                                     abi.MP_MEM_DEVICE))

    def eval_device_noalloc(self, n: int, X_ptr: int, cost_ptr: int, grad_ptr: int = 0, minc_ptr: int = 0, flags_ptr: int = 0,
                            task_of_ptr: int = 0) -> None:
        """mp_eval_noalloc: the device-pointer call whose split-mode tables live in the library's static device array
        (no hipMalloc / hipFree on the way; box model; at most MP_NOALLOC_SCRATCH_BYTES of tables)."""
        vp = C.c_void_p
        self._check(self.lib.mp_eval_noalloc(self._h, int(n), vp(X_ptr), vp(task_of_ptr or 0), vp(cost_ptr), vp(grad_ptr or 0),
                                             vp(minc_ptr or 0), vp(flags_ptr or 0)))

    # -- distance queries --
    def point_sdf(self, pts: np.ndarray, radius: float = 1e-6, box_begin: int = 0, box_count: Optional[int] = None):
        pts = np.ascontiguousarray(pts, dtype=self.np_dtype).reshape(-1, 3)
        out = np.empty(pts.shape[0], dtype=self.np_dtype)
        cnt = self.n_boxes - box_begin if box_count is None else box_count
        self._check(self.lib.mp_point_sdf(self._h, pts.shape[0], _p(pts), float(radius), box_begin, cnt, _p(out),
                                          abi.MP_MEM_HOST))
        return out

    def box_sdf(self, size, pos: np.ndarray, yaw_deg: np.ndarray, box_begin: int = 0, box_count: Optional[int] = None):
        pos = np.ascontiguousarray(pos, dtype=self.np_dtype).reshape(-1, 3)
        yaw = np.ascontiguousarray(yaw_deg, dtype=self.np_dtype).reshape(-1)
        if yaw.shape[0] != pos.shape[0]:
            raise ValueError("pos and yaw_deg must have the same length")
        sz = (C.c_double * 3)(*[float(v) for v in size])
        out = np.empty(pos.shape[0], dtype=self.np_dtype)
        cnt = self.n_boxes - box_begin if box_count is None else box_count
        self._check(self.lib.mp_box_sdf(self._h, pos.shape[0], _p(pos), _p(yaw), C.cast(sz, C.c_void_p), box_begin, cnt,
                                        _p(out), abi.MP_MEM_HOST))
        return out

    # -- device-resident CEM over many instances --
    def cem_solve(self, x0: np.ndarray, sigma0: np.ndarray, task_of: Optional[np.ndarray] = None, population_size: int = 64,
                  elite_frac: float = 0.2, max_iter: int = 90, tol: float = 1e-3, alpha: float = 0.7, min_sigma: float = 1e-3,
                  seed: Optional[int] = None, noise: Optional[np.ndarray] = None, instance_base: int = 0):
        """Batched _cem_optimize (geom/spline_opt.py:222-277): one instance per row of x0.  ``instance_base``: global index of
        row 0 when the rows are one rank's share of a longer list (the device noise streams are keyed by global index)."""
        p = self.problem
        if p is None:
            raise RuntimeError("set_problem first")
        dt = self.np_dtype
        x0 = np.ascontiguousarray(x0, dtype=dt).reshape(-1, p.dim)
        n = x0.shape[0]
        sigma0 = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma0, dtype=dt), x0.shape))
        opt = abi.MpCemOptions(int(population_size), int(max_iter), float(elite_frac), float(tol), float(alpha),
                               float(min_sigma), int(seed if seed is not None else np.random.SeedSequence().entropy & (2**63 - 1)),
                               int(instance_base))
        if noise is not None:
            noise = np.ascontiguousarray(noise, dtype=dt)
            if noise.shape != (max_iter, n, population_size, p.dim):
                raise ValueError("noise must have shape (max_iter, n_inst, population_size, D)")
        tof = None if task_of is None else np.ascontiguousarray(task_of, dtype=np.int32)
        best_x = np.empty((n, p.dim), dtype=dt)
        best_val = np.empty(n, dtype=dt)
        nit = np.empty(n, dtype=np.int32)
        minc = np.empty(n, dtype=dt)
        flags = np.empty(n, dtype=np.uint8)
        self._check(self.lib.mp_cem_solve(self._h, n, _p(tof), _p(x0), _p(sigma0), C.byref(opt), _p(noise), _p(best_x),
                                          _p(best_val), _p(nit), _p(minc), _p(flags), abi.MP_MEM_HOST))
        return {"x": best_x, "fun": best_val, "nit": nit, "min_clear": minc, "flags": flags}
