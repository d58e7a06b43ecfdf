"""Box primitive of the scene (mirrors geom/blocks.py:8-41 without the FCL object)."""
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from .utils import quat_to_rot


@dataclass
class Block:
    size: Tuple[float, float, float]          # full extents (sx, sy, sz)
    position: Tuple[float, float, float]      # centre
    quat: Tuple[float, float, float, float] = (0.0, 0.0, 0.0, 1.0)   # (x, y, z, w)
    object_id: Optional[str] = None

    def _frame(self):
        return quat_to_rot(self.quat), np.array(self.position, dtype=float), 0.5 * np.array(self.size, dtype=float)

    def contains(self, p) -> bool:
        """Closed-box membership test in the block frame (geom/blocks.py:23-29)."""
        R, T, h = self._frame()
        local = R.T @ (np.array(p, dtype=float) - T)
        return bool(np.all(np.abs(local) <= h))

    def vertices_world(self) -> np.ndarray:
        """The 8 corners in the world frame, same corner order as geom/blocks.py:31-41."""
        R, T, h = self._frame()
        signs = np.array(
            [[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]],
            dtype=float,
        )
        return (R @ (signs * h).T).T + T
