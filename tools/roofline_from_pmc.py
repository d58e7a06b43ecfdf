#!/usr/bin/env python3
"""Executed-work roofline of each kernel class from a PMC summary (tools/pmc_summary.py) and the kernel-trace stats of the
same command: executed FP64 flop = 64 x (ADD + MUL + 2 FMA + TRANS wave instructions), divided by the average launch
duration of the kernel-trace run; VALU issue utilisation = SQ_ACTIVE_INST_VALU x 4 cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024
SIMDs) of the counter run itself.
  python tools/roofline_from_pmc.py profiles/r02_config4_pmc/summary.json profiles/r02_config4_kernel_stats.csv"""
import csv, json, re, sys

PEAK64, PEAK32 = 78.6, 157.3
summ = json.load(open(sys.argv[1]))
avg_ns = {}
for r in csv.DictReader(open(sys.argv[2])):
    k = r["Name"]
    m = re.search(r"eval_kernel<\w+, \d, (true|false), \d, (true|false)", k)
    if m:
        cls = ("grad" if m.group(1) == "true" else "nograd") + ("_pre" if m.group(2) == "true" else "")
    elif "clearance_kernel" in k:
        cls = "clearance"
    else:
        continue
    avg_ns[cls] = float(r["AverageNs"])
out = {}
for cls, g in summ.items():
    if cls not in avg_ns:
        continue
    f64_insts = g["SQ_INSTS_VALU_ADD_F64"] + g["SQ_INSTS_VALU_MUL_F64"] + g["SQ_INSTS_VALU_FMA_F64"] + g["SQ_INSTS_VALU_TRANS_F64"]
    f32_insts = g["SQ_INSTS_VALU_ADD_F32"] + g["SQ_INSTS_VALU_MUL_F32"] + g["SQ_INSTS_VALU_FMA_F32"] + g["SQ_INSTS_VALU_TRANS_F32"]
    f64 = 64.0 * (f64_insts + g["SQ_INSTS_VALU_FMA_F64"])
    f32 = 64.0 * g.get("SQ_INSTS_VALU_FLOPS_FP32", 64.0 * (f32_insts + g["SQ_INSTS_VALU_FMA_F32"]) / 64.0)
    t = avg_ns[cls] * 1e-9
    cycles = g["GRBM_GUI_ACTIVE"] / 8.0
    out[cls] = {"kernel": g.get("kernel_names"), "avg_launch_ms": avg_ns[cls] * 1e-6,
                "fp64_flops_per_launch": f64, "fp64_tflops": f64 / t / 1e12, "fp64_frac_of_peak": f64 / t / 1e12 / PEAK64,
                "fp32_flops_per_launch": f32, "fp32_tflops": f32 / t / 1e12,
                "valu_insts_per_launch": g["SQ_INSTS_VALU"], "fp64_inst_share": f64_insts / g["SQ_INSTS_VALU"],
                "fp32_inst_share": f32_insts / g["SQ_INSTS_VALU"],
                "int_cvt_inst_share": (g["SQ_INSTS_VALU_INT32"] + g["SQ_INSTS_VALU_INT64"] + g["SQ_INSTS_VALU_CVT"]) / g["SQ_INSTS_VALU"],
                "salu_per_valu": g["SQ_INSTS_SALU"] / g["SQ_INSTS_VALU"],
                "valu_issue_util": g["SQ_ACTIVE_INST_VALU"] * 4 / (cycles * 1024),
                "wave_wait_frac": g["SQ_WAIT_ANY"] / g["SQ_WAVE_CYCLES"],
                "lds_bank_conflict_share_of_lds_time": g["SQ_LDS_BANK_CONFLICT"] / max(g["SQ_ACTIVE_INST_LDS"], 1.0),
                "hbm_bytes_per_launch": int((2.0 * g["FETCH_SIZE"] + g["WRITE_SIZE"]) * 1024)}
print(json.dumps(out, indent=1))
