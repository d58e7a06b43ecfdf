"""Achieved HIP-vs-oracle differences of the sweep test's legs (which scenario / method / key is sensitive)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from motion_planning_amd import sweep
from motion_planning_amd.scenarios import scenario_from_resolved
from oracle import oracle as orc
from tests import helpers as H
sweep._clock = lambda: 0.0
table = {k: v for k, v in H.load_scenarios().items() if k.startswith("step_0")}
names = sorted(table)
build = lambda n: scenario_from_resolved(table[n])
for method in ("CEM", "Powell", "CEM-POWELL"):
    hip = sweep.run_sweep(build, names, methods=(method,), trials=2, seed=7, log=None)
    o = sweep.run_sweep(build, names, methods=(method,), trials=2, seed=7, log=None, evaluator_factory=orc.OracleEvaluator)
    bh, bo = hip["benchmark"][method], o["benchmark"][method]
    print(method, "same trial", hip["hyperopt"][method]["best"]["trial"] == o["hyperopt"][method]["best"]["trial"])
    for rh, ro in zip(bh["per_scenario"], bo["per_scenario"]):
        d = {k: abs(rh[k] - ro[k]) / max(abs(ro[k]), 1e-12) for k in ("fun", "length", "min_clearance", "mean_clearance", "safety_cost", "curvature_cost", "score")}
        print(" ", rh["scenario"], rh["nit"], ro["nit"], rh["message"] == ro["message"], {k: f"{v:.1e}" for k, v in d.items()})
