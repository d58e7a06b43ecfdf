#!/usr/bin/env python3
"""Static ISA statistics of one kernel instance (investigation aid): hipcc -S of the library, then instruction mix,
SGPR-spill traffic (v_readlane / v_writelane), scratch, registers.  usage: tools/isa_stats.py [mangled-name-regex] [extra flags]"""
import collections, re, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402
pat = sys.argv[1] if len(sys.argv) > 1 else r"eval_kernelIdLi1ELb1ELi0ELb0E"
extra = sys.argv[2:]
import os
flags = [f for f in g.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + ([] if os.environ.get("NO_SCHED") == "1" else g.SCHED_FLAGS) + extra
asm = Path(tempfile.gettempdir()) / "mp_k.s"
subprocess.run(["hipcc", *flags, "--cuda-device-only", "-S", str(g.CSRC / "mp_api.hip"), "-o", str(asm)], check=True, capture_output=True)
lines = asm.read_text().splitlines()
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if m and re.search(pat, m.group(1)):
        name = m.group(1)
        j = i
        while not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i:j]
        cnt = collections.Counter()
        for l in body:
            t = l.strip()
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            cnt[t.split()[0]] += 1
        meta = {}
        for l in lines[j:j + 120]:
            mm = re.match(r"^;\s*(NumVgprs|TotalNumSgprs|ScratchSize|Occupancy|codeLenInByte|NumSgprs)\s*[:=]\s*(\d+)", l)
            if mm and mm.group(1) not in meta:
                meta[mm.group(1)] = int(mm.group(2))
        tot = sum(cnt.values())
        valu = sum(c for o, c in cnt.items() if o.startswith("v_"))
        print(name)
        print(f"  instrs {tot} valu {valu} readlane {cnt['v_readlane_b32']} writelane {cnt['v_writelane_b32']} "
              f"scratch_ld {sum(c for o, c in cnt.items() if o.startswith('scratch_load'))} scratch_st {sum(c for o, c in cnt.items() if o.startswith('scratch_store'))} "
              f"s_nop {cnt['s_nop']} v_mov {cnt['v_mov_b32_e32'] + cnt['v_mov_b64_e32']} cndmask {cnt['v_cndmask_b32_e32'] + cnt['v_cndmask_b32_e64']}  {meta}")
        i = j
    i += 1
