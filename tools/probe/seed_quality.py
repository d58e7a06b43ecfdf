#!/usr/bin/env python3
"""How good are the seed boxes of the clearance phase?  (CPU only: oracle distances + NumPy.)  For a sample of config-3 candidates:
per collision sample the exact distance to every swept box (oracle.box_sdf, one box at a time), the fp32-style AABB lower bound
the sweep tests, and for several seed choices the number of boxes the sweep would queue against the seed's exact value
(lb^2 <= U_seed; bounds do not tighten within the estimate, so it is an upper estimate of the pairs before the first drain).
usage: python tools/probe/seed_quality.py [n_boxes n_samples n_candidates]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from motion_planning_amd import _native  # noqa: E402
from motion_planning_amd.synthetic import synthetic_candidates, synthetic_problem  # noqa: E402
from oracle import oracle  # noqa: E402

B, S, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (50, 64, 256)))
prob = synthetic_problem(_native.build_basis, B, S, seed=0)
X = synthetic_candidates(prob, N, seed=2)
M = _native.build_basis(4, 3, S)
ctrl = np.asarray(prob.ctrl, float)
yawc = np.asarray(prob.yaw_ctrl, float)
P = np.empty((N, S, 3)); YW = np.empty((N, S))
for i, x in enumerate(X):
    W = ctrl.copy(); W[1] = x[0:3]; W[2] = x[3:6]
    yc = yawc.copy(); yc[1] = x[6]; yc[2] = x[7]
    P[i] = M @ W; YW[i] = M @ yc
pos = P.reshape(-1, 3); yaw = YW.reshape(-1)
boxes = np.asarray(prob.boxes, float)[:, :6]
ax, ay, az = prob.half_extents
# the wide box (table) is settled before the seeds: swept boxes only
rr = 2.0 * np.hypot(ax, ay)
wide = max((k for k in range(len(boxes)) if boxes[k, 3] >= rr and boxes[k, 4] >= rr), key=lambda k: boxes[k, 3] * boxes[k, 4], default=-1)
swept = [k for k in range(len(boxes)) if k != wide]
size = (2 * ax, 2 * ay, 2 * az)
D = np.stack([oracle.box_sdf(boxes[k:k + 1], size, pos, yaw) for k in swept], axis=1)   # (samples, boxes)
sep2 = np.where(D > 0, D * D, 0.0)     # what U holds: squared separation, 0 when penetrating
c, s = np.abs(np.cos(np.deg2rad(yaw))), np.abs(np.sin(np.deg2rad(yaw)))
ex, ey, ez = c * ax + s * ay, s * ax + c * ay, np.full_like(c, az)
bl, bh = boxes[swept, :3] - boxes[swept, 3:6], boxes[swept, :3] + boxes[swept, 3:6]
gap = np.maximum(np.maximum(bl[None] - (pos + np.stack([ex, ey, ez], 1))[:, None], (pos - np.stack([ex, ey, ez], 1))[:, None] - bh[None]), 0.0)
lb2 = (gap ** 2).sum(-1)

def pairs_for(seed_idx):
    U = sep2[np.arange(len(pos)), seed_idx]
    need = lb2 <= U[:, None] + 1e-12
    need[np.arange(len(pos)), seed_idx] = False
    return need.sum() / N, (seed_idx == D.argmin(1)).mean()

def grid_seed(cells_per_box, metric, yaw_bins=1):
    cnt = len(swept)
    lo, hi = bl.min(0), bh.max(0); ext = hi - lo
    target = min(8192.0, max(64.0, cells_per_box * cnt))
    h = np.cbrt(ext.prod() / target)
    nd = np.clip(np.ceil(ext / h), 1, 64).astype(int)
    cx = [lo[a] + (np.arange(nd[a]) + 0.5) * ext[a] / nd[a] for a in range(3)]
    G = np.stack(np.meshgrid(*cx, indexing="ij"), -1).reshape(-1, 3)
    tabs = []
    for yb in range(yaw_bins):
        if yaw_bins == 1:
            hm = np.array([0.5 * (ax + ay), 0.5 * (ax + ay), az])
        else:
            th = np.deg2rad((yb + 0.5) * 180.0 / yaw_bins)
            hm = np.array([abs(np.cos(th)) * ax + abs(np.sin(th)) * ay, abs(np.sin(th)) * ax + abs(np.cos(th)) * ay, az])
        if metric == "point":
            hm = np.zeros(3)
        q = np.abs(G[:, None] - boxes[swept, :3][None]) - (boxes[swept, 3:6] + hm)[None]
        out = np.sqrt((np.maximum(q, 0) ** 2).sum(-1))
        sd = np.where(out > 0, out, q.max(-1))
        tabs.append(sd.argmin(1))
    idx3 = np.clip(((pos - lo) * nd / ext).astype(int), 0, nd - 1)
    flat = (idx3[:, 0] * nd[1] + idx3[:, 1]) * nd[2] + idx3[:, 2]
    if yaw_bins == 1:
        return tabs[0][flat]
    yb = np.clip((np.mod(yaw, 180.0) * yaw_bins / 180.0).astype(int), 0, yaw_bins - 1)
    return np.stack(tabs, 0)[yb, flat]

print(f"B={B} S={S} N={N}: swept boxes {len(swept)}; no seed: {(lb2 <= np.inf).sum() / N:.1f} pairs per candidate (every box)")
for name, seed in [("perfect (true nearest)", D.argmin(1)),
                   ("grid 16/box, inflated (product)", grid_seed(16, "inflated")),
                   ("grid 16/box, point metric", grid_seed(16, "point")),
                   ("grid 64/box, inflated", grid_seed(64, "inflated")),
                   ("grid 16/box, inflated, 4 yaw bins", grid_seed(16, "inflated", 4)),
                   ("grid 64/box, inflated, 4 yaw bins", grid_seed(64, "inflated", 4)),
                   ("grid 64/box, inflated, 8 yaw bins", grid_seed(64, "inflated", 8))]:
    pr, hit = pairs_for(seed)
    print(f"  {name:38s}: {pr:6.1f} sweep pairs per candidate, seed is the deciding box on {100 * hit:4.1f} % of the samples")
