#!/usr/bin/env python3
"""Headline benchmark: spline-candidate cost+gradient evaluations per second.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on — "64-sample,
50-box scene"): per GPU 65,536 multi-start candidates x 64 samples x 50 axis-aligned boxes,
fp64, full 15-term box-model objective + analytic gradient + min clearance + flags.  A "step"
is one pass of the hot path over that batch (one mp_eval launch on HBM-resident inputs).
Weak scaling: every rank owns its own 65,536 candidates (sharded by index), no data-path
collective; the per-candidate (cost, min clearance, flags) of each step are all-gathered over
RCCL asynchronously straight from the kernel's output buffers (two output slots), overlapped with the next step.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python bench.py --gpus N --steps K --warmup W          # starts the N ranks itself (motion_planning_amd/launch.py)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

N_CAND = 65536
N_SAMPLES = 64
N_BOXES = 50
FP64_VALU_PEAK_TFLOPS = 78.6   # AMD MI355X datasheet, FP64 vector
FP32_VALU_PEAK_TFLOPS = 157.3
HBM_PEAK_GBPS = 8000.0


def kernel_sources_hash():
    """sha256 over the kernel sources, as tools/pmc_summary.py records it beside the counters it summarises."""
    import hashlib

    h = hashlib.sha256()
    for rel in ("motion_planning_amd/csrc/mp_kernels.hpp", "motion_planning_amd/csrc/mp_device.hpp", "motion_planning_amd/csrc/mp_api.hip"):
        h.update((ROOT / rel).read_bytes())
    return h.hexdigest()


def algorithmic_flops(S, Sp, B, C):
    """F = S'*B*f_pair + S*f_samp + 16*S*C, f_pair = 200 (box model + gradient)."""
    return Sp * B * 200 + S * 160 + 16 * S * C


def cpu_baseline(problem, X, budget_s=4.0):
    """The C oracle ("port") on the box's host cores over a bounded sample of the same workload: first on ONE
    core (the reference itself is single-threaded Python), then on `cores` threads (ctypes releases the GIL, every
    call owns its workspace) — the second figure is the reported value, the honest best-CPU number."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as orc

    orc.build()
    orc.evaluate(problem, X[:64])
    chunk = 2048

    def run(lo):
        orc.evaluate(problem, X[lo:lo + chunk], want_grad=True, want_dist=False)
        return min(chunk, X.shape[0] - lo)

    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        done += run(done % X.shape[0])
    one = done / (time.perf_counter() - t0)
    # the GPU box shows the whole machine's CPUs; a one-GPU job's share is 16
    cores = max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1))
    starts = list(range(0, X.shape[0] - chunk + 1, chunk))
    total, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(max_workers=cores) as pool:
        while time.perf_counter() - t0 < budget_s:
            total += sum(pool.map(run, starts[:4 * cores]))
    el = time.perf_counter() - t0
    # reference-shaped path (SciPy spline per call, NumPy terms, per-sample distance loop; one core like the reference)
    from oracle import numpy_path

    m, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        numpy_path.objective_batch(problem, X[m:m + 16])
        m += 16
    shaped = m / (time.perf_counter() - t0)
    return {"value": total / el, "unit": "evals/s", "cores": cores, "kind": "port", "value_one_core": one,
            "reference_shaped_numpy_one_core": shaped,
            "sample": f"{total} candidates of the same batch (64 samples x 50 boxes, fp64, cost+grad) in {el:.1f} s of "
                      f"oracle/mp_oracle.c on {cores} threads; {done} candidates in {budget_s:.0f} s on one core "
                      f"({os.cpu_count()} CPUs visible); reference-shaped NumPy/SciPy path (oracle/numpy_path.py, cost only): "
                      f"{m} candidates in 2 s on one core"}


def _rate(db, X, steps):
    db.evaluate(X)
    db.ctx.timer_start()
    for _ in range(steps):
        db.evaluate(X)
    return X.shape[0] / (db.ctx.timer_stop() / steps * 1e-3)


def workload_cases(_native, problem, Xh, np_dtype):
    """(name, problem, X) of the workload sweep: the headline shape over other scene seeds, tighter / wider candidate
    clouds, and a mostly collision-free batch (what a late CEM generation or a refined population looks like: small
    positive clearances, more pairs survive the culling bound)."""
    import torch

    from motion_planning_amd.batch import DeviceBatch, refine_gradient
    from motion_planning_amd.synthetic import synthetic_candidates, synthetic_problem

    for seed in (1, 2, 3):
        p = synthetic_problem(_native.build_basis, N_BOXES, N_SAMPLES, seed=seed)
        yield f"scene_seed_{seed}", p, synthetic_candidates(p, N_CAND, seed=200 + seed)
    rng = np.random.default_rng(300)
    eps = rng.normal(size=(N_CAND, problem.dim))
    for scale in (0.25, 1.0, 2.0):
        yield f"spread_{scale:g}_sigma", problem, problem.x0[None, :] + scale * eps * problem.sigma0[None, :]
    # a >= 90 % feasible batch: the feasible candidates of the headline cloud (~0.9 % of it), a few steps of the gradient
    # refinement (more steps trade the last centimetres of clearance for path length and curvature: with BASE_CONFIG's
    # weights the optimum of this cluttered scene sits at a clearance just below the required one), the survivors that
    # are still feasible, and slightly jittered copies of them to refill the batch
    db = DeviceBatch(problem, N_CAND, dtype=np_dtype, want_grad=True)
    Xd = torch.from_numpy(np.ascontiguousarray(Xh, dtype=np_dtype)).cuda()
    raw_ok = (db.evaluate(Xd)["flags"] & 2) != 0
    surv = Xd[raw_ok].contiguous()
    if surv.shape[0] >= 64:
        ref = refine_gradient(db, surv, iters=5)
        ok = (ref["flags"] & 2) != 0
        if int(ok.sum()) >= 64:
            surv = ref["x"][ok]
    db.close()
    if surv.shape[0] >= 64:
        surv = surv.cpu().numpy().astype(np.float64)
        pick = rng.integers(0, surv.shape[0], size=N_CAND)
        yield "refined_mostly_feasible", problem, surv[pick] + 0.01 * rng.normal(size=(N_CAND, problem.dim)) * problem.sigma0[None, :]


def workload_sweep(_native, problem, Xh, np_dtype, steps):
    """Throughput of the headline shape (65,536 x 64 samples x 50 boxes, cost + gradient) over workload_cases().  Exact
    pairs per candidate come from tools/workload_stats.py (-DMP_STATS build) when its summary is committed under profiles/."""
    import torch

    from motion_planning_amd.batch import DeviceBatch

    stats = {}
    sfile = max((ROOT / "profiles").glob("r*_workload_stats.json"), default=ROOT / "profiles" / "none.json")
    if sfile.exists():
        try:
            stats = json.loads(sfile.read_text())
        except Exception:
            stats = {}
    rows = []
    for name, prob, Xn in workload_cases(_native, problem, Xh, np_dtype):
        Xd = torch.from_numpy(np.ascontiguousarray(Xn, dtype=np_dtype)).cuda()
        db = DeviceBatch(prob, Xd.shape[0], dtype=np_dtype, want_grad=True)
        rate = _rate(db, Xd, steps)
        torch.cuda.synchronize()
        fl = db.flags.cpu().numpy()
        row = {"case": name, "evals_per_s": rate, "colliding_fraction": float((fl & 1).mean()),
               "feasible_fraction": float(((fl & 2) != 0).mean())}
        if name in stats:
            row["exact_pairs_per_candidate"] = stats[name].get("exact_pairs_per_candidate")
        rows.append(row)
        db.close()
    vals = sorted(r["evals_per_s"] for r in rows)
    return {"cases": rows, "min_evals_per_s": vals[0], "median_evals_per_s": vals[len(vals) // 2], "max_evals_per_s": vals[-1],
            "shape": "65,536 candidates x 64 samples x 50 boxes, fp64 cost + gradient, one MI355X"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.1 s of timed GPU work.  A step is 0.5 ms, and the first ~10 ms after the set-up phase run ~7 % slower
    # (20 steps after 3 warm-up steps: 0.517 ms per step; 200 after 20: 0.480 ms, the kernel's own duration)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-mode", action="store_true",
                    help="only the W + K launches of the headline kernel (what tools/pmc_passes.sh and the rocprofv3 --stats run "
                         "profile): no side metrics, no CPU baseline")
    ap.add_argument("--sustained-s", type=float, default=1.0, help="length of the sustained leg after the headline (N = 1)")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N > 1: the (cost, min clearance, flags) of this many consecutive steps go into ONE all-gather (a ring of "
                         "result buffers per rank; fewer, larger collectives: a gather per 0.4 ms step costs ~8 %% of the step)")
    ap.add_argument("--settle-s", type=float, default=0.5,
                    help="seconds of back-to-back launches of the same step on every rank BEFORE the W warm-up steps, so that the "
                         "timed K steps run at settled clocks whatever W and K are (the first ~10 ms after an idle phase run ~7 %% "
                         "slower); not part of the timed region, reported as extras.settle_launches")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a fresh child process, BEFORE anything
    # here touches the GPU (motion_planning_amd/launch.py), relay its JSON line and exit with its code.  Fewer than N
    # visible devices is an error, never a quiet one-rank run that prints n_gpus: 1.
    from motion_planning_amd import launch

    one_device = os.environ.get("MP_BENCH_ONE_DEVICE") == "1"
    if args.gpus > 1 and not launch.under_launcher():
        sys.exit(launch.self_launch(str(Path(__file__).resolve()), sys.argv[1:], args.gpus, one_device=one_device))
    if launch.under_launcher() and int(os.environ["WORLD_SIZE"]) != args.gpus:
        if int(os.environ.get("RANK", "0")) == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ['WORLD_SIZE']} ranks; "
                  f"give both the same number", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist

    from motion_planning_amd import _native
    from motion_planning_amd.batch import DeviceBatch
    from motion_planning_amd.synthetic import synthetic_candidates, synthetic_problem

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (one-GPU box): MP_BENCH_ONE_DEVICE=1 puts every rank on cuda:0, MP_BENCH_BACKEND=gloo
    # swaps RCCL for gloo; the driver's runs use neither
    if one_device:
        local_rank = 0
    backend = os.environ.get("MP_BENCH_BACKEND", "nccl")
    # MP_BENCH_FORCE_DIST=1 under torchrun with ONE rank drives the whole collective path (RCCL communicator, async gathers,
    # barrier) on a one-GPU box
    distributed = world > 1 or os.environ.get("MP_BENCH_FORCE_DIST") == "1"
    if not one_device and torch.cuda.device_count() <= local_rank:
        print(f"bench.py: rank {rank} has no GPU (LOCAL_RANK={local_rank}, {torch.cuda.device_count()} visible)", file=sys.stderr)
        sys.exit(2)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    np_dtype = np.float64 if args.dtype == "f64" else np.float32
    tdtype = torch.float64 if args.dtype == "f64" else torch.float32

    problem = synthetic_problem(_native.build_basis, N_BOXES, N_SAMPLES, seed=0)
    Xh = synthetic_candidates(problem, N_CAND, seed=100 + rank, dtype=np_dtype)   # this rank's index shard
    X = torch.from_numpy(Xh).cuda()
    # two rings of G result buffers: the collective of one group of G steps reads ring g % 2 while the next G steps compute
    # into the other one
    G = max(1, args.gather_every) if distributed else 1
    dbs = [DeviceBatch(problem, N_CAND, device=local_rank, dtype=np_dtype, want_grad=True, result_slots=G)
           for _ in range(2 if distributed else 1)]
    db = dbs[0]
    # collectives run on device tensors over RCCL; the gloo rehearsal (one-GPU box) stages them through the host
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    if distributed:
        gather_buf = [torch.empty(world * G * db.slot_bytes, dtype=torch.uint8, device=coll_dev) for _ in range(2)]

    sent = [0, 0]   # per ring: first entry of the current group not yet handed to a collective

    def step(i, pending, n_steps):
        """Step i of a run of n_steps.  Collectives: one per group of G steps, except that the run's last group is cut
        after its second-to-last step, so that only ONE step's results (1.1 MB per rank) are still to be exchanged when
        the last kernel ends and the bulk of the last group travels under that kernel."""
        grp, slot = (i // G) % len(dbs), i % G
        cur = dbs[grp]
        if distributed and slot == 0:
            sent[grp] = 0
            if len(pending) >= 2:
                for w in pending.pop(0)[1]:   # the collectives that read this ring two groups ago: long finished as a rule,
                    if not w.is_completed():   # and then the compute stream needs no wait packet in front of the next launch
                        w.wait()
        out = cur.evaluate(X, slot=slot)
        if distributed and (slot == G - 1 or i >= n_steps - 2):
            # the kernels' own output buffers go into the collective (cost, min clearance and flags of a step share one
            # allocation, the steps of a group are consecutive in it): no packing kernel, the gather runs on RCCL's stream
            # while the next steps compute
            f, k = sent[grp], slot + 1
            res = cur.results_ring[f:k].reshape(-1)
            if coll_dev != "cuda":
                res = res.cpu()
            dst = gather_buf[grp][world * f * db.slot_bytes: world * k * db.slot_bytes]
            w = dist.all_gather_into_tensor(dst, res, async_op=True)
            if pending and pending[-1][0] == (i // G):
                pending[-1][1].append(w)
            else:
                pending.append(((i // G), [w]))
            sent[grp] = k
            last_gather.update(grp=grp, first=f, end=k)
        return out

    last_gather = {}

    def sync():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    # first use of every collective of the job (communicator set-up, first barrier: tens of ms on RCCL) happens here, not
    # between the warm-up and the timed steps, where the idle GPU would drop its clocks
    if distributed:
        dist.barrier()
        dist.all_gather_into_tensor(gather_buf[0], dbs[0].results_ring.reshape(-1) if coll_dev == "cuda" else dbs[0].results_ring.reshape(-1).cpu())
        t_warm = torch.zeros(1, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t_warm, op=dist.ReduceOp.MAX)
        dist.barrier()
    # clock settling (see --settle-s): plain launches of the step's kernel, no collectives
    settle_launches = 0
    if args.settle_s > 0 and not args.profile_mode:
        t_end = time.perf_counter() + args.settle_s
        while time.perf_counter() < t_end:
            for j in range(64):   # every context and every ring entry is touched here, not first inside the timed region
                dbs[(j // G) % len(dbs)].evaluate(X, slot=j % G)
            torch.cuda.synchronize()
            settle_launches += 64
    pend = []
    for i in range(args.warmup):
        step(i, pend, args.warmup)
    for _, ws in pend:
        for w in ws:
            w.wait()
    sync()
    pend = []
    # HIP events on the launch stream bracket the same K launches the wall clock times (roofline: average launch duration)
    for d in dbs:
        d.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    t0 = time.perf_counter()
    db.ctx.timer_start()
    for i in range(args.steps):
        step(i, pend, args.steps)
    kern_ms = db.ctx.timer_stop() / max(args.steps, 1)   # waits for the last launch
    t_k = time.perf_counter()
    for _, ws in pend:
        for w in ws:
            w.wait()
    torch.cuda.synchronize()
    t_g = time.perf_counter()
    sync()
    elapsed = time.perf_counter() - t0
    tail_ms = {"last_launch_done": (t_k - t0) * 1e3, "gathers_done": (t_g - t0) * 1e3, "barrier_done": elapsed * 1e3}
    gather_ok = None
    if distributed:
        # this rank's own rows of the last collective are the bytes its kernels wrote
        grp, f, k = last_gather["grp"], last_gather["first"], last_gather["end"]
        region = gather_buf[grp][world * f * db.slot_bytes: world * k * db.slot_bytes]
        mine = region[rank * (k - f) * db.slot_bytes: (rank + 1) * (k - f) * db.slot_bytes]
        gather_ok = bool(torch.equal(mine.cpu(), dbs[grp].results_ring[f:k].reshape(-1).cpu()))
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # roofline of the dominant kernel: kern_ms = HIP-event time of the timed region / K (all K launches are on one stream,
    # back to back; with N > 1 the gathers run on RCCL's stream beside them).  `achieved` is EXECUTED FP64 work: the typed
    # VALU instruction counters of gfx950 (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64, one count per wave instruction = 64
    # lanes, an FMA = 2 flop) collected by rocprofv3 on this same command (`bench.py --profile-mode`, tools/pmc_passes.sh ->
    # profiles/r<NN>_pmc[_f32]/summary.json; the counts per launch are a property of the kernel and the batch, not of the clock),
    # divided by the launch duration measured live here.
    bf_flops = algorithmic_flops(N_SAMPLES, N_SAMPLES, N_BOXES, problem.n_ctrl) * N_CAND
    peak = FP64_VALU_PEAK_TFLOPS if args.dtype == "f64" else FP32_VALU_PEAK_TFLOPS
    es = 8 if args.dtype == "f64" else 4
    alg_bytes = (es * problem.dim + es * (problem.dim + 2) + 1) * N_CAND
    traffic, executed, achieved, frac = None, None, None, None
    # the newest committed counter set of this dtype whose provenance names the kernel sources of THIS tree (the counts per
    # launch are a property of the kernels and the batch; a summary taken on other sources would report a stale utilisation,
    # so it is refused and achieved / frac stay null)
    pmc_note = "no counter summary under profiles/"
    for pfile in sorted((ROOT / "profiles").glob("r*_pmc" + ("" if args.dtype == "f64" else "_f32") + "/summary.json"), reverse=True):
        try:
            summ = json.loads(pfile.read_text())
            g = summ["grad"]
            prov = summ.get("_provenance", {}).get("kernel_sources_sha256")
            if prov != kernel_sources_hash():
                pmc_note = f"{pfile.relative_to(ROOT)} was taken on other kernel sources: refused"
                continue
            f64_insts = g["SQ_INSTS_VALU_ADD_F64"] + g["SQ_INSTS_VALU_MUL_F64"] + g["SQ_INSTS_VALU_FMA_F64"] + g["SQ_INSTS_VALU_TRANS_F64"]
            f32_insts = g["SQ_INSTS_VALU_ADD_F32"] + g["SQ_INSTS_VALU_MUL_F32"] + g["SQ_INSTS_VALU_FMA_F32"] + g["SQ_INSTS_VALU_TRANS_F32"]
            f64_flops = 64.0 * (f64_insts + g["SQ_INSTS_VALU_FMA_F64"])          # an FMA counts twice
            f32_flops = 64.0 * g.get("SQ_INSTS_VALU_FLOPS_FP32", 0.0)            # per-instruction flops, packed operations included
            own_flops = f64_flops if args.dtype == "f64" else f32_flops
            achieved = own_flops / (kern_ms * 1e-3) / 1e12
            frac = achieved / peak
            # every wave64 VALU instruction occupies its SIMD's 16-lane ALU for 4 cycles: issue utilisation of the vector
            # ALUs = instructions x 4 cycles / (SIMDs x cycles of the launch)
            cycles = g["GRBM_GUI_ACTIVE"] / 8.0       # the counter sums the 8 XCDs
            executed = {"fp64_flops_per_launch": f64_flops, "fp64_flops_per_eval": f64_flops / N_CAND,
                        "fp32_flops_per_launch": f32_flops, "fp32_flops_per_eval": f32_flops / N_CAND,
                        "valu_insts_per_eval": g["SQ_INSTS_VALU"] / N_CAND, "fp64_insts_per_eval": f64_insts / N_CAND,
                        "fp32_insts_per_eval": f32_insts / N_CAND,
                        "fp64_add_mul_fma_per_eval": [g["SQ_INSTS_VALU_ADD_F64"] / N_CAND, g["SQ_INSTS_VALU_MUL_F64"] / N_CAND,
                                                      g["SQ_INSTS_VALU_FMA_F64"] / N_CAND],
                        "int_cvt_insts_per_eval": (g["SQ_INSTS_VALU_INT32"] + g["SQ_INSTS_VALU_INT64"] + g["SQ_INSTS_VALU_CVT"]) / N_CAND,
                        "salu_insts_per_eval": g["SQ_INSTS_SALU"] / N_CAND, "lds_insts_per_eval": g["SQ_INSTS_LDS"] / N_CAND,
                        "valu_issue_util": g["SQ_ACTIVE_INST_VALU"] * 4 / (cycles * 1024),
                        "wave_wait_frac": g["SQ_WAIT_ANY"] / g["SQ_WAVE_CYCLES"],
                        "launch": ", ".join(g.get("kernel_names", [])) + ", 65,536 candidates (this command's batch)",
                        "source": f"{pfile.relative_to(ROOT)} (rocprofv3 --pmc passes over `bench.py --profile-mode`, tools/prof_round.sh)"}
            # HBM bytes per launch: FETCH_SIZE (KiB, x2 on gfx950 per MI355X_MICROARCH.md) + WRITE_SIZE (KiB), separate passes
            traffic = int((2.0 * g["FETCH_SIZE"] + g["WRITE_SIZE"]) * 1024)
            pmc_note = None
            break
        except Exception as exc:
            traffic, executed, achieved, frac = None, None, None, None
            pmc_note = f"{pfile.name}: {exc!r}"

    extras = {"timed_region_ms": tail_ms, "gather_own_rows_match": gather_ok, "settle_launches": settle_launches, "settle_note": "untimed launches of the same step before the warm-up steps (clock settling, --settle-s)"}
    if args.profile_mode:
        if rank == 0:
            print(json.dumps({"profile_mode": True, "kernel_ms": kern_ms, "ms_per_step": elapsed / args.steps * 1e3, "steps": args.steps}))
        for d in dbs:
            d.close()
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return
    if rank == 0 and world == 1:
        # sustained leg: >= 1 s of back-to-back launches of the same step (the headline's K steps are a few ms after an
        # idle phase; this is the settled-clock figure and what a GPU-busy sampler can see)
        n_sus = max(args.steps, int(args.sustained_s / max(kern_ms * 1e-3, 1e-6)))
        db.ctx.timer_start()
        for i in range(n_sus):
            db.evaluate(X)
        sus_ms = db.ctx.timer_stop() / n_sus
        extras["sustained_ms_per_step"] = sus_ms
        extras["sustained_steps"] = n_sus
        extras["sustained_evals_per_s"] = N_CAND / (sus_ms * 1e-3)
        db_c = DeviceBatch(problem, N_CAND, dtype=np_dtype, want_grad=False)
        db_c.evaluate(X)
        db_c.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        db_c.ctx.timer_start()
        for _ in range(args.steps):
            db_c.evaluate(X)
        extras["cost_only_evals_per_s"] = N_CAND / (db_c.ctx.timer_stop() / args.steps * 1e-3)
        db_c.close()
        other = np.float32 if args.dtype == "f64" else np.float64
        db_o = DeviceBatch(problem, N_CAND, dtype=other, want_grad=True)
        Xo = torch.from_numpy(Xh.astype(other)).cuda()
        db_o.evaluate(Xo)
        db_o.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        db_o.ctx.timer_start()
        for _ in range(args.steps):
            db_o.evaluate(Xo)
        extras[("fp32" if args.dtype == "f64" else "fp64") + "_cost_grad_evals_per_s"] = N_CAND / (db_o.ctx.timer_stop() / args.steps * 1e-3)
        db_o.close()
        if not args.no_cpu_baseline:
            # independent batches pipelined over two streams (two contexts, two output slots): the next batch's
            # ramp-up fills the tail of the previous one.  Side figure only; `value` and the roofline above are
            # single-stream.  Skipped with --no-cpu-baseline (the rocprofv3 run), whose per-kernel average
            # duration would otherwise mix overlapped launches of the same kernel in.
            db2 = DeviceBatch(problem, N_CAND, dtype=np_dtype, want_grad=True)
            pair, strs = [db, db2], [torch.cuda.Stream(), torch.cuda.Stream()]
            for rep in range(2):
                torch.cuda.synchronize()
                tp = time.perf_counter()
                for i in range(2 * args.steps):
                    with torch.cuda.stream(strs[i % 2]):
                        pair[i % 2].evaluate(X)
                torch.cuda.synchronize()
                tp = time.perf_counter() - tp
            extras["two_streams_evals_per_s"] = 2 * args.steps * N_CAND / tp
            db2.close()
        db.evaluate(X)
        torch.cuda.synchronize()
        fl = db.flags.cpu().numpy()
        extras["colliding_fraction"] = float((fl & 1).mean())
        extras["feasible_fraction"] = float(((fl & 2) != 0).mean())
        if not args.no_cpu_baseline:
            extras["workload_sweep"] = workload_sweep(_native, problem, Xh, np_dtype, max(20, args.steps // 4))

    # second headline metric (BASELINE config 5, bounded): scenarios solved per second, every rank its own scenarios
    # x 32 restarts of device-resident CEM to the reference stopping rule; not part of `value`
    sc = {}
    try:
        from motion_planning_amd.batch import solve_scenarios_cem
        from motion_planning_amd.synthetic import wall_scenario_problems

        n_sc = int(os.environ.get("MP_BENCH_SCENARIOS", "512"))
        probs = wall_scenario_problems(_native.build_basis, n_sc, seed=2 + 7919 * rank)
        sync()
        t1 = time.perf_counter()
        res = solve_scenarios_cem(probs, restarts=32, device=local_rank, dtype=np.float64, population_size=64, max_iter=90,
                                  seed=1000 * rank)
        cnt = torch.tensor([float(res["solved"].sum()), float(n_sc)], dtype=torch.float64, device=coll_dev)
        sync()
        el5 = time.perf_counter() - t1
        if distributed:
            dist.all_reduce(cnt)
            t5 = torch.tensor([el5], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t5, op=dist.ReduceOp.MAX)
            el5 = float(t5.item())
        sc = {"scenarios_solved_per_s": float(cnt[0].item()) / el5, "scenarios": int(cnt[1].item()), "restarts": 32,
              "solved_fraction": float(cnt[0].item() / cnt[1].item()), "wall_s": el5,
              "workload": "BASELINE config 5 (bounded): generated wall-building scenarios, CEM pop 64 to the reference stop rule, S=101"}
    except Exception as exc:   # the headline line must survive a failure of the side metric
        sc = {"error": repr(exc)}

    if rank == 0:
        extras["config5"] = sc
        line = {
            "metric": "spline cost+grad evals/sec (node)",
            "value": world * N_CAND * args.steps / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "BASELINE config 3: 65,536 multi-start spline candidates x 64 samples x 50 axis-aligned "
                                   "boxes per GPU, 15-term box-model cost + gradient + min clearance + flags",
                       "candidates_per_gpu": N_CAND, "n_samples": N_SAMPLES, "n_boxes": N_BOXES, "dim": problem.dim,
                       "parallelism": f"index-sharded x{world}, async all-gather of (cost,min_clear,flags)"
                                      + (f", one collective per {G} steps" if distributed else "")},
            "roofline": {"bound": "fp64_valu" if args.dtype == "f64" else "fp32_valu", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": frac, "traffic": traffic,
                         "kernel_ms": kern_ms, "executed": executed, "counters_refused": pmc_note,
                         "algorithmic_bytes_per_eval": alg_bytes // N_CAND,
                         "hbm_algorithmic_GBps": alg_bytes / (kern_ms * 1e-3) / 1e9, "hbm_peak_GBps": HBM_PEAK_GBPS,
                         "bruteforce_flops_per_eval": bf_flops // N_CAND,
                         "bruteforce_equiv_tflops": bf_flops / (kern_ms * 1e-3) / 1e12,
                         "algorithmic_speedup_vs_bruteforce": bf_flops / (kern_ms * 1e-3) / 1e12 / peak,
                         "note": "bound = FP64 vector ALU (SURVEY 8d: no MFMA-shaped work, 145 B/eval of HBM traffic); peak = "
                                 "datasheet FP64 vector rate.  achieved / frac = EXECUTED FP64 flop/s from the typed VALU counters "
                                 "(fp32 launches: the FP32 flop counter against the FP32 vector peak; null when no counter summary of these kernel sources is committed, see counters_refused); valu_issue_util = share of vector-ALU "
                                 "issue cycles in use (all instruction types).  The kernel culls ~92 % of the (sample, box) pair "
                                 "evaluations with a conservative bound, so SURVEY's brute-force flop count F is not executed: "
                                 "bruteforce_equiv_tflops = F x evals / time and algorithmic_speedup_vs_bruteforce = that / peak "
                                 "(> 1 means faster than an ideal brute-force kernel) are throughput figures, not utilisation"},
            "extras": extras,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(problem, Xh.astype(np.float64))
        print(json.dumps(line))
    for d in dbs:
        d.close()
    if distributed:
        dist.barrier()   # rank 0 prints and times the CPU baseline last; nobody tears the group down early
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
