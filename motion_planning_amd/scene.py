"""Scene container with the reference's duck-typed surface (geom/scene.py), without FCL.

Host-side bookkeeping (ids, stacking helpers) is plain Python.  The two distance queries,
``signed_distance`` (geom/scene.py:50-68) and ``signed_distance_block`` (:70-122), are
answered by the HIP library's batched distance kernels (mp_point_sdf / mp_box_sdf) with the
exact-geometry semantics documented in DESIGN.md; the ``*_many`` variants expose the batch.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .blocks import Block
from .problem import as_boxes10, scene_to_boxes
from .utils import boxes_in_frame, quat_is_yaw_only, quat_to_rot, yaw_quat_to_deg

IdLike = Union[int, str]

# face -> (normal axis, sign, tangential axes) in the base block's local frame
# (+x right, -x left, +y front, -y back, +z top, -z bottom; geom/scene.py:172-234)
_FACES: Dict[str, Tuple[int, int, Tuple[int, int]]] = {
    "top": (2, +1, (0, 1)),
    "bottom": (2, -1, (0, 1)),
    "front": (1, +1, (0, 2)),
    "back": (1, -1, (0, 2)),
    "right": (0, +1, (1, 2)),
    "left": (0, -1, (1, 2)),
}


class Scene:
    def __init__(self, device: int = 0):
        self.blocks: List[Block] = []
        self._id_to_index: Dict[str, int] = {}
        self._auto_id_counter = 0
        self._device = device
        self._ctx = None
        self._ctx_key = None
        self._ctx_frame = None    # second context: tilted-mover queries (the scene re-expressed in the mover frame)

    # ---- ids -------------------------------------------------------------------------
    def _claim_id(self, wanted: Optional[str]) -> str:
        """The id a new block gets: the caller's (must be unused) or the next automatic ``obj_<k>``."""
        if wanted is not None:
            if wanted in self._id_to_index:
                raise ValueError(f"object_id '{wanted}' already exists in scene.")
            return wanted
        self._auto_id_counter += 1
        return f"obj_{self._auto_id_counter - 1}"

    def _resolve(self, ref: IdLike) -> int:
        """Index of a block given its list index or its object_id."""
        if isinstance(ref, int):
            if ref < 0 or ref >= len(self.blocks):
                raise IndexError(f"Block index {ref} out of range.")
            return ref
        try:
            return self._id_to_index[ref]
        except KeyError:
            raise KeyError(f"object_id '{ref}' not found.") from None

    def add_block(self, size, position, quat=(0.0, 0.0, 0.0, 1.0), object_id: Optional[str] = None) -> str:
        """Append a block; returns its object_id (geom/scene.py:35-40)."""
        oid = self._claim_id(object_id)
        self._id_to_index[oid] = len(self.blocks)
        self.blocks.append(Block(size=size, position=position, quat=quat, object_id=oid))
        self._ctx_key = None
        return oid

    def get_block(self, id_or_index: IdLike) -> Block:
        return self.blocks[self._resolve(id_or_index)]

    # ---- distance queries (native) -------------------------------------------------------
    def _native(self, ignore_ids: Optional[Sequence[str]] = None):
        """The context with this scene resident.  The key is the CONTENT of the box table (size, position, quaternion of every
        block that is not ignored), so a block mutated in place is uploaded again; building the table is O(blocks) host work,
        far below the cost of a query."""
        from . import _native

        rows = np.ascontiguousarray(scene_to_boxes(self, ignore_ids))
        key = (rows.shape, rows.tobytes())
        if self._ctx is None:
            self._ctx = _native.Context(self._device, np.float64)
        if self._ctx_key != key:
            self._ctx.set_scene(rows)
            self._ctx_key = key
        return self._ctx

    def signed_distance_many(self, pts, point_radius: float = 1e-6) -> np.ndarray:
        pts = np.asarray(pts, dtype=float).reshape(-1, 3)
        if not self.blocks:
            return np.full(pts.shape[0], np.inf)
        return self._native().point_sdf(pts, point_radius)

    def signed_distance(self, p, point_radius: float = 1e-6) -> float:
        """Signed distance from a point to the union of blocks: > 0 outside, < 0 inside."""
        return float(self.signed_distance_many(np.asarray(p, dtype=float).reshape(1, 3), point_radius)[0])

    def signed_distance_block_many(self, size, positions, yaw_deg, ignore_ids: Optional[Sequence[str]] = None) -> np.ndarray:
        positions = np.asarray(positions, dtype=float).reshape(-1, 3)
        yaw = np.broadcast_to(np.asarray(yaw_deg, dtype=float).reshape(-1), (positions.shape[0],))
        ctx = self._native(ignore_ids)
        if ctx.n_boxes == 0:
            return np.full(positions.shape[0], np.inf)
        return ctx.box_sdf(size, positions, np.ascontiguousarray(yaw))

    def signed_distance_block(self, size, position, quat=(0.0, 0.0, 0.0, 1.0), ignore_ids: Optional[List[str]] = None) -> float:
        """Signed distance from a moving box to the scene blocks (geom/scene.py:70-122):
        positive = separated, 0 = touching, negative = minus the penetration depth.

        A yaw-only ``quat`` (what the planner produces) is one query against the resident scene.  Any other rotation
        is answered in the moving box's own frame: the static boxes are re-expressed there on the host
        (``utils.boxes_in_frame``) and the box, unrotated at the origin, is queried against them."""
        if quat_is_yaw_only(quat):
            return float(self.signed_distance_block_many(size, np.asarray(position, dtype=float).reshape(1, 3),
                                                         [yaw_quat_to_deg(quat)], ignore_ids)[0])
        rows = as_boxes10(scene_to_boxes(self, ignore_ids))
        if rows.shape[0] == 0:
            return float("inf")
        from . import _native

        if self._ctx_frame is None:      # a second context: the resident scene of the yaw-only queries stays where it is
            self._ctx_frame = _native.Context(self._device, np.float64)
        self._ctx_frame.set_scene(boxes_in_frame(rows, position, quat))
        return float(self._ctx_frame.box_sdf(size, np.zeros((1, 3)), np.zeros(1))[0])

    def sample_sdf_grid(self, bounds, dims):
        """Point SDF on a regular grid (geom/scene.py:157-170), one batched query."""
        (xmin, xmax), (ymin, ymax), (zmin, zmax) = bounds
        nx, ny, nz = dims
        xs, ys, zs = np.linspace(xmin, xmax, nx), np.linspace(ymin, ymax, ny), np.linspace(zmin, zmax, nz)
        gx, gy, gz = np.meshgrid(xs, ys, zs, indexing="ij")
        pts = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1)
        return (xs, ys, zs), self.signed_distance_many(pts).reshape(nx, ny, nz)

    # ---- face-based stacking helpers (geom/scene.py:172-277) ---------------------------------
    def _axes_center_half_extents(self, b: Block):
        return quat_to_rot(b.quat), np.asarray(b.position, dtype=float), 0.5 * np.asarray(b.size, dtype=float)

    def get_stack_point_on_face(self, base: IdLike, new_size, face: str, gap: float = 0.0,
                                tangential_offset: Tuple[float, float] = (0.0, 0.0)) -> np.ndarray:
        """Centre of a block of ``new_size`` placed against ``face`` of ``base`` (geom/scene.py:185-245)."""
        b = self.blocks[self._resolve(base)]
        R, c, h_base = self._axes_center_half_extents(b)
        h_new = 0.5 * np.asarray(new_size, dtype=float)
        key = face.lower()
        if key not in _FACES:
            raise ValueError("face must be one of: top, bottom, front, back, right, left")
        axis, sign, (ta, tb) = _FACES[key]
        sep = h_base[axis] + h_new[axis] + gap
        u_off, v_off = tangential_offset
        return c + (R[:, axis] * sign) * sep + R[:, ta] * u_off + R[:, tb] * v_off

    def get_top_point(self, base, new_size, gap=0.0, xy_offset=(0.0, 0.0)):
        return self.get_stack_point_on_face(base, new_size, "top", gap, xy_offset)

    def get_bottom_point(self, base, new_size, gap=0.0, xy_offset=(0.0, 0.0)):
        return self.get_stack_point_on_face(base, new_size, "bottom", gap, xy_offset)

    def get_front_point(self, base, new_size, gap=0.0, xz_offset=(0.0, 0.0)):
        return self.get_stack_point_on_face(base, new_size, "front", gap, xz_offset)

    def get_back_point(self, base, new_size, gap=0.0, xz_offset=(0.0, 0.0)):
        return self.get_stack_point_on_face(base, new_size, "back", gap, xz_offset)

    def get_right_point(self, base, new_size, gap=0.0, yz_offset=(0.0, 0.0)):
        return self.get_stack_point_on_face(base, new_size, "right", gap, yz_offset)

    def get_left_point(self, base, new_size, gap=0.0, yz_offset=(0.0, 0.0)):
        return self.get_stack_point_on_face(base, new_size, "left", gap, yz_offset)

    def stack_on(self, base, size, xy_offset=(0.0, 0.0), quat=(0.0, 0.0, 0.0, 1.0), gap=0.0, object_id=None) -> str:
        pos = self.get_top_point(base, size, gap=gap, xy_offset=xy_offset)
        return self.add_block(size=size, position=tuple(pos.tolist()), quat=quat, object_id=object_id)

    def stack_on_face(self, base, size, face, tangential_offset=(0.0, 0.0), quat=(0.0, 0.0, 0.0, 1.0), gap=0.0,
                      object_id=None) -> str:
        pos = self.get_stack_point_on_face(base, size, face, gap=gap, tangential_offset=tangential_offset)
        return self.add_block(size=size, position=tuple(pos.tolist()), quat=quat, object_id=object_id)
