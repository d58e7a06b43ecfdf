"""MI355X-native batched B-spline trajectory evaluation (drop-in for the hot path of
Geryyy/motion_planning).  The heavy lifting is in csrc/ (HIP kernels behind the C ABI of
include/mp_api.h); this package is the Python host that mirrors the reference's planner
surface.  Importing the package does not load the native library; the first call that
needs it does, and fails loudly when it is missing."""

__all__ = ["spline_opt", "scene", "blocks", "utils", "problem", "batch", "scenarios", "planning"]
__version__ = "0.1.0"
