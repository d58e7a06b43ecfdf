"""ctypes mirror of include/mp_api.h (structs and constants only, no library loading).

Kept separate from ``_native`` so that host logic and the CPU-side tests can build
problem descriptors without a GPU or the HIP library being present.
"""
from __future__ import annotations

import ctypes as C

MP_API_VERSION = 2
MP_MAX_CTRL = 16
MP_MAX_SAMPLES = 1024
MP_NTERMS = 15

MP_OK = 0
MP_E_ARG = -1
MP_E_STATE = -2
MP_E_HIP = -3
MP_E_UNSUPPORTED = -4
MP_E_NOMEM = -5

MP_F64 = 0
MP_F32 = 1
MP_MODEL_POINT = 0
MP_MODEL_BOX = 1
MP_MEM_HOST = 0
MP_MEM_DEVICE = 1

MP_FLAG_COLLISION = 1
MP_FLAG_FEASIBLE = 2

# enum mp_term, order of the weighted sum geom/spline_opt.py:605-621
TERM_NAMES = (
    "length",
    "curvature_cost",
    "yaw_smoothness_cost",
    "safety_cost",
    "preferred_safety_cost",
    "approach_rebound_cost",
    "goal_clearance_cost",
    "goal_clearance_target_cost",
    "approach_clearance_cost",
    "approach_collision_cost",
    "via_deviation_cost",
    "yaw_deviation_cost",
    "yaw_monotonic_cost",
    "yaw_schedule_cost",
    "goal_approach_normal_cost",
)
(T_LEN, T_CURV, T_YAW_SMOOTH, T_SAFE, T_SAFE_PREF, T_REBOUND, T_GOAL_CLEAR, T_GOAL_TARGET,
 T_APPROACH_CLEAR, T_APPROACH_COL, T_VIA_DEV, T_YAW_DEV, T_YAW_MONO, T_YAW_SCHED,
 T_GOAL_NORMAL) = range(MP_NTERMS)

TERM_MASK_ALL = (1 << MP_NTERMS) - 1
TERM_MASK_3TERM = (1 << T_LEN) | (1 << T_CURV) | (1 << T_SAFE)


class MpConfig(C.Structure):
    _fields_ = [
        ("n_samples", C.c_int32),
        ("n_ctrl", C.c_int32),
        ("first_free", C.c_int32),
        ("n_free", C.c_int32),
        ("n_yaw_ctrl", C.c_int32),
        ("n_yaw_free", C.c_int32),
        ("model", C.c_int32),
        ("n_safe", C.c_int32),
        ("last_if_not_close", C.c_int32),
        ("keep_n", C.c_int32),
        ("use_pref", C.c_int32),
        ("tail_start", C.c_int32),
        ("use_goal_target", C.c_int32),
        ("goal_tail_n", C.c_int32),
        ("term_mask", C.c_uint32),
        ("reserved0", C.c_int32),
        ("weights", C.c_double * MP_NTERMS),
        ("required_clearance", C.c_double),
        ("preferred_clearance", C.c_double),
        ("approach_target", C.c_double),
        ("goal_clearance_target", C.c_double),
        ("point_radius", C.c_double),
        ("half_extents", C.c_double * 3),
        ("idx_safe", C.POINTER(C.c_int32)),
        ("approach_mask", C.POINTER(C.c_uint8)),
        ("basis", C.POINTER(C.c_double)),
        ("basis_yaw", C.POINTER(C.c_double)),
        ("t_sched", C.POINTER(C.c_double)),
    ]


class MpTask(C.Structure):
    _fields_ = [
        ("ctrl", (C.c_double * 3) * MP_MAX_CTRL),
        ("yaw_ctrl", C.c_double * MP_MAX_CTRL),
        ("goal_normal", C.c_double * 3),
        ("has_goal_normal", C.c_int32),
        ("yaw_increasing", C.c_int32),
        ("box_begin", C.c_int32),
        ("box_count", C.c_int32),
    ]


class MpCemOptions(C.Structure):
    _fields_ = [
        ("population_size", C.c_int32),
        ("max_iter", C.c_int32),
        ("elite_frac", C.c_double),
        ("tol", C.c_double),
        ("alpha", C.c_double),
        ("min_sigma", C.c_double),
        ("seed", C.c_uint64),
        ("instance_base", C.c_uint64),
    ]
