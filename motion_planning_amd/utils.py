"""Quaternion helper of the reference's geometry layer (geom/utils.py:3-14)."""
import numpy as np


def quat_to_rot(q):
    """Rotation matrix of a quaternion given as [x, y, z, w]."""
    x, y, z, w = (float(v) for v in q)
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
        ]
    )


def quat_is_yaw_only(q) -> bool:
    return float(q[0]) == 0.0 and float(q[1]) == 0.0


def yaw_quat_to_deg(q) -> float:
    """Yaw angle (degrees) of a pure-yaw quaternion (0, 0, sin h, cos h)."""
    return float(np.degrees(2.0 * np.arctan2(float(q[2]), float(q[3]))))
