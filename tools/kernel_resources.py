#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel of the library (hipcc -S with the build flags), written to
profiles/r02_kernel_resources.txt.  Runs on the CPU (cross-compile), ~1 min."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402

flags = [f for f in g.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + g.SCHED_FLAGS
with tempfile.TemporaryDirectory() as tmp:
    asm = Path(tmp) / "k.s"
    subprocess.run(["hipcc", *flags, "--cuda-device-only", "-S", str(g.CSRC / "mp_api.hip"), "-o", str(asm)], check=True,
                   capture_output=True)
    lines = asm.read_text().splitlines()
keys = {"NumVgprs": "v", "TotalNumSgprs": "s", "ScratchSize": "sc", "Occupancy": "o", "codeLenInByte": "c"}
rows, cur, vals = [], None, {}
for line in lines:
    m = re.match(r"^(_Z\w+):\s*;\s*@", line)
    if m:
        if cur and "o" in vals:
            rows.append((cur, vals))
        cur, vals = m.group(1), {}
        continue
    m = re.match(r"^;\s*(\w+)(?:\s*=|:)\s*(\d+)", line)
    if m and m.group(1) in keys and keys[m.group(1)] not in vals:
        vals[keys[m.group(1)]] = int(m.group(2))
if cur and "o" in vals:
    rows.append((cur, vals))
names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.strip().split("\n")
out = ["# Resource usage of every kernel in libmp_hip.so (hipcc -S of motion_planning_amd/csrc/mp_api.hip with the build flags of",
       "# __graft_entry__.py; gfx950; tools/kernel_resources.py).  waves/SIMD = what the register budget allows (LDS may allow",
       "# fewer, DESIGN.md 4).  eval_kernel<T, MODEL(0 point/1 box), GRAD, ROT(0 axis-aligned/1 yaw/2 tilted), PRE(split mode)>", "",
       f"{'kernel':58s} {'VGPR':>5s} {'SGPR':>5s} {'scratch B':>9s} {'waves/SIMD':>10s} {'code B':>7s}"]
for (_, v), d in zip(rows, names):
    d = re.sub(r"\(.*$", "", d).replace("void ", "").replace("mp::", "")
    out.append(f"{d:58s} {v.get('v', -1):5d} {v.get('s', -1):5d} {v.get('sc', -1):9d} {v.get('o', -1):10d} {v.get('c', -1):7d}")
(ROOT / "profiles" / (sys.argv[1] if len(sys.argv) > 1 else "r03_kernel_resources.txt")).write_text("\n".join(out) + "\n")
print("\n".join(out))
