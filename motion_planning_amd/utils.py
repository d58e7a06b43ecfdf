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


def quat_mul(a, b):
    """Hamilton product a (x) b of quaternions given as [x, y, z, w] (rotation b first, then a)."""
    ax, ay, az, aw = (float(v) for v in a)
    bx, by, bz, bw = (float(v) for v in b)
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def boxes_in_frame(rows10: np.ndarray, position, quat) -> np.ndarray:
    """(B,10) box rows expressed in the frame of a body at ``position`` with rotation ``quat``: a rigid motion leaves
    every distance unchanged, so a query with an arbitrarily rotated moving box becomes a query with an unrotated one
    at the origin against these rows."""
    rows = np.array(rows10, dtype=float)
    q = np.asarray(quat, dtype=float)
    q = q / np.linalg.norm(q)
    qi = np.array([-q[0], -q[1], -q[2], q[3]])
    Rt = quat_to_rot(q).T
    rows[:, :3] = (rows[:, :3] - np.asarray(position, dtype=float)[None, :]) @ Rt.T
    for i in range(rows.shape[0]):
        r = quat_mul(qi, rows[i, 6:10])
        rows[i, 6:10] = r / np.linalg.norm(r)
    return rows
