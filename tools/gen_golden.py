"""Generate tests/golden/* from the REAL reference module.

Runs only in the build container: it loads /root/reference/geom/spline_opt.py stand-alone
by path (the module needs only NumPy + SciPy; `import geom` itself fails on the missing
python-fcl).  The reference's `scene` argument is duck-typed, so it is handed an
OracleScene whose two distance methods are the oracle's exact-geometry queries: every
number downstream of the clearance profile — spline sampling, the 15 terms and their
windows, the optimiser drivers — is then computed by reference code and committed as data.
The fixtures hold inputs and outputs only, never reference source text.

Usage: python tools/gen_golden.py   (writes tests/golden/*.npz / *.json)
"""
from __future__ import annotations

import importlib.util
import json
import sys
from pathlib import Path

import numpy as np
import scipy
import yaml

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")
OUT = ROOT / "tests" / "golden"

from motion_planning_amd.blocks import Block  # noqa: E402
from motion_planning_amd.configs import BASE_CONFIG  # noqa: E402
from motion_planning_amd.scene import Scene  # noqa: E402
from motion_planning_amd.scenarios import resolve_goal, parse_goal_normals  # noqa: E402
from motion_planning_amd.synthetic import random_scene  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def load_ref():
    spec = importlib.util.spec_from_file_location("ref_spline_opt", str(REF / "geom" / "spline_opt.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class RecordingScene(orc.OracleScene):
    """OracleScene that remembers every distance it returned."""

    def __init__(self, blocks):
        super().__init__(blocks)
        self.log = []

    def signed_distance(self, p, point_radius=1e-6):
        v = super().signed_distance(p, point_radius)
        self.log.append(v)
        return v

    def signed_distance_block(self, size, position, quat=(0.0, 0.0, 0.0, 1.0), ignore_ids=None):
        v = super().signed_distance_block(size, position, quat, ignore_ids)
        self.log.append(v)
        return v


# reference tuple order (spline_opt.py:622-641) -> mp_term order (:605-621)
REF_TO_TERM = [1, 2, 4, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]


def resolve_yaml_scenarios():
    """Read the reference's two scenario files and resolve goals with the product's own loader
    logic (the reference's loader imports geom -> fcl and cannot run here)."""
    out = {}
    for fname in ("scenarios.yaml", "generated_scenarios.yaml"):
        payload = yaml.safe_load((REF / "example" / fname).read_text())
        base = tuple(float(v) for v in payload.get("defaults", {}).get("base_size", [0.6, 0.9, 0.6]))
        for name, cfg in payload["scenarios"].items():
            scene = Scene()
            blocks = []
            for item in cfg.get("blocks", []):
                size = tuple(float(v) for v in item.get("size", base))
                pos = tuple(float(v) for v in item["position"])
                scene.add_block(size=size, position=pos, object_id=str(item["id"]))
                blocks.append({"id": str(item["id"]), "size": list(size), "position": list(pos)})
            mv = cfg["moving_block"]
            msize = tuple(float(v) for v in mv.get("size", base))
            goal = resolve_goal(scene, msize, mv["goal"])
            out[name] = {
                "file": fname, "blocks": blocks, "start": [float(v) for v in mv["start"]], "goal": list(goal),
                "moving_block_size": list(msize), "start_yaw_deg": float(mv.get("start_yaw_deg", 0.0)),
                "goal_yaw_deg": float(mv.get("goal_yaw_deg", 0.0)),
                "goal_normals": [list(n) for n in parse_goal_normals(mv, name)],
            }
    return out


def blocks_of(entry):
    return [Block(tuple(b["size"]), tuple(b["position"]), object_id=b["id"]) for b in entry["blocks"]]


def capture_objective(ref, fn_name, scene, args, kwargs):
    """Run a reference planner with its solver replaced, to get hold of objective_single."""
    got = {}

    def fake(objective_single, x0, sigma0, method, options=None):
        got.update(obj=objective_single, x0=np.array(x0, float), sigma0=np.array(sigma0, float))
        return {"x": x0, "success": True, "message": "", "nit": 0, "fun": float(objective_single(x0)[0])}

    saved = ref._solve_optimizer
    ref._solve_optimizer = fake
    try:
        getattr(ref, fn_name)(scene, *args, **kwargs)
    finally:
        ref._solve_optimizer = saved
    return got


def jsonable(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, np.ndarray):
            out[k] = v.tolist()
        elif isinstance(v, (np.floating, np.integer)):
            out[k] = v.item()
        else:
            out[k] = v
    return out


def gen_g1(ref, rng):
    data = {}
    for C, S in [(4, 64), (4, 101), (5, 121), (6, 256), (8, 101), (4, 3), (7, 64)]:
        W = rng.normal(size=(C, 4)) * np.array([1.0, 1.0, 1.0, 40.0])
        data[f"W_{C}_{S}"] = W
        data[f"Q_{C}_{S}"] = ref.build_cubic_bspline(W)(np.linspace(0.0, 1.0, S))
    for n in (2, 3, 4, 5):
        y = rng.normal(size=n) * 30.0
        data[f"y_{n}"] = y
        data[f"yq_{n}"] = ref.build_scalar_bspline(y)(np.linspace(0.0, 1.0, 77))
    np.savez(OUT / "g1_spline.npz", **data)


def gen_g2(ref, rng):
    data = {}
    for i, S in enumerate([2, 3, 5, 64, 101]):
        P = np.cumsum(rng.normal(size=(S, 3)) * 0.05, axis=0)
        if i == 3:
            P[10:14] = P[10]  # stationary stretch: speed below eps, zero-length segments
        yaw = np.cumsum(rng.normal(size=S) * 3.0)
        normals = rng.normal(size=(2, 3))
        data[f"P_{S}"] = P
        data[f"yaw_{S}"] = yaw
        data[f"normals_{S}"] = normals
        data[f"out_{S}"] = np.array([
            ref.path_length(P), ref.curvature_cost(P), ref.yaw_smoothness_cost(yaw), ref.mean_turn_angle_deg(P),
            ref._goal_approach_alignment_cost(P, normals, 0.12),
            ref._goal_approach_alignment_cost(P, np.zeros((1, 3)), 0.12),
        ])
    yaws = np.concatenate([rng.uniform(-720, 720, 200), [0.0, 90.0, -90.0, 180.0, 45.0, 360.0, 1e-9]])
    data["quat_yaw"] = yaws
    data["quat"] = np.array([ref.yaw_deg_to_quat(y) for y in yaws])
    for n in (0, 1, 2, 5):
        data[f"via_init_{n}"] = ref._default_via_initialization(np.array([-1.0, 0.5, 2.0]), np.array([2.0, -1.0, 0.3]), n)
    np.savez(OUT / "g2_terms.npz", **data)


def gen_g3(ref, rng, scenarios):
    cases = []
    arrays = {}

    def add_case(name, fn_name, blocks, args, kwargs, n=12, scales=(0.25, 1.0)):
        scene = RecordingScene(blocks)
        cap = capture_objective(ref, fn_name, scene, args, kwargs)
        D = cap["x0"].size
        X = np.vstack([cap["x0"][None, :]] + [cap["x0"] + cap["sigma0"] * rng.normal(size=(n // len(scales), D)) * sc
                                                for sc in scales])
        outs, dists = [], []
        for x in X:
            scene.log = []
            r = cap["obj"](x)
            if len(r) == 4:   # 3-term objective: (j, len, curv, safe)
                row = np.zeros(16)
                row[0], row[1], row[2], row[4] = r[0], r[1], r[2], r[3]
            else:
                row = np.array([r[0]] + [r[i] for i in REF_TO_TERM], dtype=float)
            outs.append(row)
            dists.append(np.array(scene.log, dtype=float))
        width = max(len(d) for d in dists)
        dd = np.full((len(dists), width), np.nan)
        for i, d in enumerate(dists):
            dd[i, : len(d)] = d
        arrays[f"{name}__X"] = X
        arrays[f"{name}__out"] = np.array(outs)
        arrays[f"{name}__dist"] = dd
        arrays[f"{name}__x0"] = cap["x0"]
        arrays[f"{name}__sigma0"] = cap["sigma0"]
        kw = {k: (np.asarray(v).tolist() if isinstance(v, np.ndarray) else v) for k, v in kwargs.items()}
        cases.append({"name": name, "fn": fn_name, "args": [np.asarray(a).tolist() for a in args], "kwargs": kw,
                      "blocks": [{"id": b.object_id, "size": list(map(float, b.size)), "position": list(map(float, b.position))}
                                 for b in blocks]})

    def path_kwargs(sc, **over):
        kw = dict(BASE_CONFIG)
        kw.update(moving_block_size=tuple(sc["moving_block_size"]), start_yaw_deg=sc["start_yaw_deg"],
                  goal_yaw_deg=sc["goal_yaw_deg"], goal_approach_normals=np.array(sc["goal_normals"], dtype=float))
        kw.update(over)
        return kw

    for name, sc in scenarios.items():
        add_case(f"base_{name}", "optimize_bspline_path", blocks_of(sc), (sc["start"], sc["goal"]), path_kwargs(sc))
    fr = scenarios["front"]
    bl = blocks_of(fr)
    sg = (fr["start"], fr["goal"])
    add_case("front_sub3", "optimize_bspline_path", bl, sg, path_kwargs(fr, collision_check_subsample=3))
    add_case("front_sub4_s64", "optimize_bspline_path", bl, sg, path_kwargs(fr, collision_check_subsample=4, n_samples_curve=64))
    add_case("front_sep_yaw1", "optimize_bspline_path", bl, sg, path_kwargs(fr, combined_4d=False, n_yaw_vias=1))
    add_case("front_sep_yaw0", "optimize_bspline_path", bl, sg, path_kwargs(fr, combined_4d=False, n_yaw_vias=0))
    add_case("front_sep_yaw3", "optimize_bspline_path", bl, sg, path_kwargs(fr, combined_4d=False, n_yaw_vias=3))
    add_case("front_norelax", "optimize_bspline_path", bl, sg, path_kwargs(fr, relax_preferred_final_fraction=0.0))
    add_case("front_vias3_s121", "optimize_bspline_path", bl, sg, path_kwargs(fr, n_vias=3, n_yaw_vias=3, n_samples_curve=121))
    add_case("front_vias5_s256", "optimize_bspline_path", bl, sg, path_kwargs(fr, n_vias=5, n_yaw_vias=5, n_samples_curve=256), n=6)
    add_case("front_point", "optimize_bspline_path", bl, sg, path_kwargs(fr, moving_block_size=None))
    add_case("front_tool", "optimize_bspline_path", bl, sg,
             path_kwargs(fr, moving_block_size=None, tool_half_extents=(0.2, 0.1, 0.15)))
    add_case("front_nonormals", "optimize_bspline_path", bl, sg, path_kwargs(fr, goal_approach_normals=None))
    add_case("front_ignore_wall", "optimize_bspline_path", bl, sg, path_kwargs(fr, collision_ignore_ids=["wall", "cube_top"]))
    add_case("front_notarget", "optimize_bspline_path", bl, sg,
             path_kwargs(fr, goal_clearance_target=None, approach_only_clearance=None, preferred_safety_margin=None,
                         safety_margin=0.03))
    add_case("front_defaults", "optimize_bspline_path", bl, sg,
             dict(moving_block_size=tuple(fr["moving_block_size"]), goal_yaw_deg=90.0, n_vias=2))
    add_case("front_tinywindow", "optimize_bspline_path", bl, sg,
             path_kwargs(fr, n_samples_curve=9, contact_window_fraction=0.999, approach_fraction=0.01,
                         goal_approach_window_fraction=0.01))
    for seed in (0, 1):
        start, goal = (-1.5, -1.5, 1.5), (1.5, 1.5, 0.4)
        sc = random_scene(50, seed, (0.1, 0.6), start, goal)
        kw = dict(BASE_CONFIG, n_samples_curve=64, moving_block_size=(0.9, 0.6, 0.6), start_yaw_deg=0.0,
                  goal_yaw_deg=90.0, goal_approach_normals=np.array([[0.0, 0.0, 1.0]]))
        add_case(f"synth50_seed{seed}", "optimize_bspline_path", sc.blocks, (start, goal), kw)
    via = (-0.2, 0.5, 1.2)
    goal3 = (0.0, 0.25, 0.8)
    add_case("vias_point", "optimize_bspline_with_vias", bl, (fr["start"], via, goal3), dict(n_additional_vias=2))
    add_case("vias_point_sub7", "optimize_bspline_with_vias", bl, (fr["start"], via, goal3),
             dict(n_additional_vias=3, collision_check_subsample=7))
    add_case("vias_box_sub5", "optimize_bspline_with_vias", bl, (fr["start"], via, goal3),
             dict(n_additional_vias=1, collision_check_subsample=5, tool_half_extents=(0.3, 0.2, 0.2), n_samples_curve=64))
    add_case("vias_stationary_end", "optimize_bspline_with_vias", bl, (fr["start"], via, (0.0, 0.25, 0.8)),
             dict(n_additional_vias=2, collision_check_subsample=7, n_samples_curve=121), scales=(1e-12,), n=2)
    np.savez_compressed(OUT / "g3_objective.npz", **arrays)
    (OUT / "g3_cases.json").write_text(json.dumps(cases, indent=1))


def gen_g4(ref, rng):
    data = {}

    def analytic(X):
        return np.sum((X - np.array([0.3, -0.2, 0.5, 1.0])) ** 2, axis=1) + 0.1 * np.sin(5.0 * X[:, 0])

    for seed in (0, 1, 7):
        pops = []

        def obj(X, pops=pops):
            pops.append(X.copy())
            return analytic(X)

        res = ref._cem_optimize(obj, x0=np.zeros(4), sigma0=np.array([0.5, 0.5, 1.0, 0.2]), population_size=24,
                                elite_frac=0.25, max_iter=40, tol=1e-3, alpha=0.6, min_sigma=1e-3, seed=seed)
        data[f"x_{seed}"] = res["x"]
        data[f"fun_{seed}"] = np.array(res["fun"])
        data[f"nit_{seed}"] = np.array(res["nit"])
        data[f"popmean_{seed}"] = np.array([p.mean(axis=0) for p in pops])
        data[f"popstd_{seed}"] = np.array([p.std(axis=0) for p in pops])
    np.savez(OUT / "g4_cem.npz", **data)


def gen_g5(ref, scenarios):
    runs = []
    numeric_keys = None

    def record(name, fn_name, sc_name, args, kwargs, extra=None):
        sc = scenarios[sc_name]
        scene = orc.OracleScene(blocks_of(sc))
        res = getattr(ref, fn_name)(scene, *args, **kwargs)
        info = res[-1]
        vias = np.array(res[1:-1] if fn_name.endswith("two_vias") else res[1], dtype=float)
        rec = {"name": name, "fn": fn_name, "scenario": sc_name, "args": [np.asarray(a).tolist() for a in args],
               "kwargs": {k: (np.asarray(v).tolist() if isinstance(v, np.ndarray) else v) for k, v in kwargs.items()},
               "vias": vias.tolist(), "info": {}}
        for k, v in info.items():
            if callable(v):
                continue
            if isinstance(v, np.ndarray):
                rec["info"][k] = v.tolist()
            elif isinstance(v, (bool, np.bool_)):
                rec["info"][k] = bool(v)
            elif isinstance(v, (int, float, np.floating, np.integer)):
                rec["info"][k] = float(v) if not isinstance(v, (int, np.integer)) else int(v)
            elif v is None or isinstance(v, str):
                rec["info"][k] = v
        S = res[0]
        rec["curve_at_u"] = np.asarray(S(np.linspace(0.0, 1.0, 7))).tolist()
        runs.append(rec)
        print(f"  g5 {name}: fun={info['fun']:.6f} nit={info['nit']}")

    def pk(sc_name, **over):
        sc = scenarios[sc_name]
        kw = dict(BASE_CONFIG)
        kw.update(moving_block_size=tuple(sc["moving_block_size"]), start_yaw_deg=sc["start_yaw_deg"],
                  goal_yaw_deg=sc["goal_yaw_deg"], goal_approach_normals=np.array(sc["goal_normals"], dtype=float))
        kw.update(over)
        return kw

    fr = scenarios["front"]
    record("front_powell_demo", "optimize_bspline_path", "front", (fr["start"], fr["goal"]),
           pk("front", method="Powell", options={"maxiter": 160, "xtol": 1e-3, "ftol": 1e-3}))
    record("front_nelder", "optimize_bspline_path", "front", (fr["start"], fr["goal"]),
           pk("front", method="Nelder-Mead", options={"maxiter": 120}))
    record("front_cem", "optimize_bspline_path", "front", (fr["start"], fr["goal"]),
           pk("front", method="CEM", options={"population_size": 32, "max_iter": 25, "seed": 3}))
    record("front_hybrid", "optimize_bspline_path", "front", (fr["start"], fr["goal"]),
           pk("front", method="CEM-POWELL", options={"cem": {"population_size": 24, "max_iter": 12, "seed": 5},
                                                      "powell": {"maxiter": 20}}))
    s3 = scenarios["step_03_third_on_top"]
    record("step03_cem", "optimize_bspline_path", "step_03_third_on_top", (s3["start"], s3["goal"]),
           pk("step_03_third_on_top", method="CEM", options={"population_size": 32, "max_iter": 20, "seed": 11}))
    via, goal3 = (-0.2, 0.5, 1.2), (0.0, 0.25, 0.8)
    record("vias_cem", "optimize_bspline_with_vias", "front", (fr["start"], via, goal3),
           dict(n_additional_vias=2, method="CEM", options={"population_size": 32, "max_iter": 20, "seed": 2}))
    record("two_vias_nelder", "optimize_bspline_two_vias", "front", (fr["start"], via, goal3), dict(n_samples_curve=61))
    (OUT / "g5_solve.json").write_text(json.dumps(runs, indent=1))


def _extract_functions(path, names, namespace):
    """Compile the named top-level functions (and the named top-level assignments) of a reference source file into
    `namespace` WITHOUT importing the module: example/benchmark_planners.py and geom/planning.py import geom -> fcl at
    module level, but the bodies needed here use NumPy only.  Nothing of the source text is kept."""
    import ast

    tree = ast.parse(Path(path).read_text())
    keep = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            keep.append(node)
        elif isinstance(node, (ast.Assign, ast.AnnAssign)):
            targets = node.targets if isinstance(node, ast.Assign) else [node.target]
            if any(isinstance(t, ast.Name) and t.id in names for t in targets):
                keep.append(node)
    found = {n.name if isinstance(n, ast.FunctionDef) else (n.targets[0].id if isinstance(n, ast.Assign) else n.target.id) for n in keep}
    missing = set(names) - found
    if missing:
        raise RuntimeError(f"{path}: {sorted(missing)} not found")
    exec(compile(ast.Module(body=keep, type_ignores=[]), str(path), "exec"), namespace)
    return namespace


def gen_g6():
    """The two host helpers whose counterparts were only checked by eye: the sweep's random-search draws and
    score (example/benchmark_planners.py:30-117) and the placement helper (geom/planning.py:6-95), run from the reference's
    own function bodies."""
    from typing import Any, Dict, Tuple

    ns = {"np": np, "Dict": Dict, "Any": Any, "Tuple": Tuple}
    _extract_functions(REF / "example" / "benchmark_planners.py", {"BASE_CONFIG", "_sample_strategy_config", "_scenario_score"}, ns)
    draws = []
    for method, seed in (("Powell", 7), ("CEM", 10007), ("CEM-POWELL", 20007), ("hybrid", 123), ("powell", 99), ("cem", 5)):
        rng = np.random.default_rng(seed)
        trials = []
        for _ in range(12):     # one generator over the trials of a method, as in _hyperopt (benchmark_planners.py:239-269)
            cfg, opts = ns["_sample_strategy_config"](method, rng)
            # the config is BASE_CONFIG with the drawn keys replaced: only what differs is stored
            trials.append({"config_diff": {k: v for k, v in cfg.items() if ns["BASE_CONFIG"].get(k) != v},
                           "config_keys": len(cfg), "options": opts})
        draws.append({"method": method, "seed": seed, "trials": trials})
    scores = []
    rs = np.random.default_rng(61)
    for _ in range(24):
        info = {"fun": float(rs.uniform(0.5, 400.0)), "min_clearance": float(rs.uniform(-0.2, 0.2)), "success": bool(rs.random() < 0.7)}
        if rs.random() < 0.15:
            del info["min_clearance"]
        if rs.random() < 0.15:
            del info["success"]
        rt = float(rs.uniform(0.0, 30.0))
        scores.append({"info": info, "runtime_s": rt, "score": ns["_scenario_score"](info, rt)})
    (OUT / "g6_sweep_draws.json").write_text(json.dumps({"base_config": ns["BASE_CONFIG"], "draws": draws, "scores": scores}, indent=1))

    uns = {"np": np}
    _extract_functions(REF / "geom" / "utils.py", {"quat_to_rot"}, uns)
    pns = {"np": np, "Tuple": Tuple, "Scene": object, "quat_to_rot": uns["quat_to_rot"]}
    _extract_functions(REF / "geom" / "planning.py", {"goal_and_via_for_placement"}, pns)

    class _Blk:
        def __init__(self, size, position, quat):
            self.size, self.position, self.quat = size, position, quat

    class _Scene:
        def __init__(self, blk):
            self.blk = blk

        def get_block(self, _id):
            return self.blk

    rp = np.random.default_rng(62)
    cases = []
    dirs = ["front", "back", "left", "right", "top", "Front", "TOP"]
    for i in range(28):
        size = [float(v) for v in rp.uniform(0.2, 1.2, 3)]
        pos = [float(v) for v in rp.uniform(-1.5, 1.5, 3)]
        if i % 3 == 0:
            quat = [0.0, 0.0, 0.0, 1.0]
        elif i % 3 == 1:
            h = 0.5 * float(rp.uniform(-np.pi, np.pi))
            quat = [0.0, 0.0, float(np.sin(h)), float(np.cos(h))]
        else:
            q = rp.normal(size=4)
            quat = [float(v) for v in q / np.linalg.norm(q)]
        kw = {"gap": float(rp.uniform(0.0, 0.05)), "lateral_offset": [float(v) for v in rp.uniform(-0.3, 0.3, 2)],
              "interlock_x": float(rp.uniform(-0.2, 0.2)), "via_shift_along_normal": float(rp.uniform(0.0, 0.2)),
              "via_offset_local": [float(v) for v in rp.uniform(-0.3, 0.3, 3)]}
        if i < 5:
            kw = {}        # the defaults
        d = dirs[i % len(dirs)]
        goal, via = pns["goal_and_via_for_placement"](_Scene(_Blk(size, pos, quat)), "b", d,
                                                       **{k: (tuple(v) if isinstance(v, list) else v) for k, v in kw.items()})
        cases.append({"size": size, "position": pos, "quat": quat, "direction": d, "kwargs": kw,
                      "goal": [float(v) for v in goal], "via": [float(v) for v in via]})
    err = None
    try:
        pns["goal_and_via_for_placement"](_Scene(_Blk([1, 1, 1], [0, 0, 0], [0, 0, 0, 1])), "b", "bottom")
    except ValueError as exc:
        err = str(exc)
    (OUT / "g6_placement.json").write_text(json.dumps({"cases": cases, "bad_direction_message": err}, indent=1))


def gen_payload():
    """The parsed content of the reference's example/generated_scenarios.yaml (a data file, the output of its
    generate_scenarios.py): pins motion_planning_amd.synthetic.generate_step_scenarios."""
    payload = yaml.safe_load((REF / "example" / "generated_scenarios.yaml").read_text())
    (OUT / "generated_payload.json").write_text(json.dumps(payload, indent=1))


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    if "--payload-only" in sys.argv:
        gen_payload()
        return
    if "--g6-only" in sys.argv:
        gen_g6()
        return
    ref = load_ref()
    rng = np.random.default_rng(20260220)
    scenarios = resolve_yaml_scenarios()
    meta = {"numpy": np.__version__, "scipy": scipy.__version__, "reference": "Geryyy/motion_planning @ 2026-02-20",
            "generator": "tools/gen_golden.py"}
    (OUT / "scenarios.json").write_text(json.dumps({"meta": meta, "scenarios": scenarios}, indent=1))
    gen_g1(ref, rng)
    gen_g2(ref, rng)
    gen_g3(ref, rng, scenarios)
    gen_g4(ref, rng)
    gen_g5(ref, scenarios)
    gen_g6()
    gen_payload()
    (OUT / "META.json").write_text(json.dumps(meta, indent=1))
    print("wrote", sorted(p.name for p in OUT.iterdir()))


if __name__ == "__main__":
    main()
