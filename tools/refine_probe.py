import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from motion_planning_amd import _native
from motion_planning_amd.batch import DeviceBatch, refine_gradient
from motion_planning_amd.synthetic import synthetic_candidates, synthetic_problem
prob = synthetic_problem(_native.build_basis, 50, 64, 0)
n = 4096
X = torch.from_numpy(prob.x0 + 0.2 * prob.sigma0 * np.random.default_rng(2).normal(size=(n, prob.dim))).cuda()
db = DeviceBatch(prob, n, want_grad=True, want_terms=True)
o = {k: v.clone() for k, v in db.evaluate(X).items()}
print("start cost median", o["cost"].median().item(), "terms median", o["terms"].median(dim=0).values.cpu().numpy().round(3))
print("weights", prob.weights)
print("grad median abs", o["grad"].abs().median(dim=0).values.cpu().numpy())
for it in (10, 30, 100):
    r = refine_gradient(db, X, iters=it)
    ratio = (r["cost"] / o["cost"])
    print(it, "median ratio", ratio.median().item(), "min", ratio.min().item(), "frac<0.9", (ratio < 0.9).double().mean().item())
print("---- trace")
x = X.clone()
out = db.evaluate(x)
cost = out["cost"].clone(); grad = out["grad"].clone()
scale = torch.as_tensor(prob.sigma0, dtype=x.dtype, device=x.device)
step = torch.full_like(cost, 0.05)
for it in range(40):
    gs = grad * scale[None, :]
    gnorm = gs.norm(dim=1, keepdim=True).clamp_min(1e-30)
    trial = (x - step[:, None] * scale[None, :] * gs / gnorm).contiguous()
    o2 = db.evaluate(trial)
    better = o2["cost"] < cost
    x = torch.where(better[:, None], trial, x)
    grad = torch.where(better[:, None], o2["grad"], grad)
    cost = torch.where(better, o2["cost"], cost)
    step = torch.where(better, step * 1.3, step * 0.5)
    if it % 4 == 0:
        chk = db.evaluate(x.contiguous())["cost"]
        print(it, "median cost", cost.median().item(), "accepted", better.double().mean().item(), "median step", step.median().item(),
              "recheck max abs diff", (chk - cost).abs().max().item())
