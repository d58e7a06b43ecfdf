"""Placement helper of the reference's public surface (``geom/planning.py:6-95``, exported from
``geom/__init__.py:4``): goal and via point for putting the next block against a face of a placed one.

Host-only geometry.  The per-direction conventions below are the reference's, kept as they are
(including that 'front' measures along the block's local y axis with the full y extent and 'back'
along minus local z with the full x extent): callers tuned their offsets against them."""
from __future__ import annotations

from typing import Tuple

import numpy as np

from .utils import quat_to_rot

# direction -> (normal column of R, sign, tangent columns (u, v), extent along the normal as (axis, factor))
_FACES = {
    "front": (1, +1.0, (0, 2), (1, 2.0)),
    "back": (2, -1.0, (0, 2), (0, 2.0)),
    "left": (0, +1.0, (1, 2), (0, 2.0)),
    "right": (0, -1.0, (1, 2), (0, 2.0)),
    "top": (2, +1.0, (0, 1), (2, 1.0)),
}


def goal_and_via_for_placement(
    scene,
    base_id,
    direction: str,
    gap: float = 0.01,
    lateral_offset: Tuple[float, float] = (0.0, 0.0),
    interlock_x: float = 0.0,
    via_shift_along_normal: float = 0.05,
    via_offset_local: Tuple[float, float, float] = (0.0, 0.0, 0.25),
):
    """Returns ``(goal (3,), via (3,))`` in world coordinates.

    ``scene`` needs ``get_block(id_or_index)``; the block's quaternion may be any rotation.
    ``interlock_x`` is added to the first lateral offset for 'top' placements only (:83-85)."""
    blk = scene.get_block(base_id)
    R = quat_to_rot(blk.quat)
    centre = np.asarray(blk.position, dtype=float)
    half = 0.5 * np.asarray(blk.size, dtype=float)
    key = direction.lower()
    if key not in _FACES:
        raise ValueError("direction must be one of {'front','back','top'}")
    ncol, sign, (ucol, vcol), (ext_axis, ext_factor) = _FACES[key]
    n_hat = sign * R[:, ncol]
    u_off, v_off = lateral_offset
    if key == "top":
        u_off = lateral_offset[0] + interlock_x
    reach = ext_factor * half[ext_axis]
    goal = centre + n_hat * (reach + gap) + R[:, ucol] * u_off + R[:, vcol] * v_off
    via = goal + (R @ np.asarray(via_offset_local, dtype=float)) + n_hat * via_shift_along_normal
    return goal, via
