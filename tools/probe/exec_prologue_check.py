#!/usr/bin/env python3
"""Scan a gfx950 assembly listing (hipcc --cuda-device-only -S) for vector instructions that sit at the head of a
basic block AHEAD of the `s_or_b64 exec, exec, ...` that re-opens the lanes of the region ending there.

Why: round 2's 4-waves-per-SIMD build of the fp64 gradient kernel returned gradients without their M^T g part.  The
listing shows the cause: the register allocator split the live range of `lane & 7` and put the copy
(`v_mov_b32 v23, v51`) at the top of the join block of a `for (s = lane; s < S; s += 64)` loop, in front of the
`s_or_b64 exec, exec, s[14:15]` that restores the lanes.  A loop of that shape leaves with exec == 0, so the copy wrote
no lane, the later copy back handed the gather loop a stale constant, and its trip test failed for every S <= 256.
(profiles/r03_forced128_*.txt.)  A vector write in that position is only right when the value is
dead in the lanes that are switched off.  Program code at the tail of an `if` body that closes an inner region sits in
the same position and is right, so the check reports the sure case: the block is the fall-through of the loop's
`s_cbranch_execnz` back edge (exec == 0 there, the instruction writes no lane at all).

usage: exec_prologue_check.py listing.s [...]      exit status 1 when one sits in a loop-exit block (-v lists the others too)
"""
import re
import sys

LABEL = re.compile(r"^(\.LBB\d+_\d+|_Z\w+):")
EXEC_OR = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")
# vector instructions that write lanes under exec (lane moves v_readlane/v_writelane/v_readfirstlane ignore exec)
VEC = re.compile(r"^\s*(v_(?!readlane|writelane|readfirstlane)\w+|ds_\w+|global_\w+|scratch_\w+|buffer_\w+|flat_\w+)\b")
# register-to-register copies and spill traffic: what the allocator inserts on its own (program code at the tail of an
# `if` body that closes an inner region is also found ahead of an exec restore, and is right there)
COPY = re.compile(r"^\s*(v_mov_b32_e32\s+v\d+,\s*v\d+|v_mov_b64_e32\s+v\[\d+:\d+\],\s*v\[\d+:\d+\]|v_accvgpr_\w+|scratch_(load|store)\w+.*Folded)")
ENDS = re.compile(r"^\s*(s_cbranch\w*|s_branch|s_endpgm|s_and_saveexec_b64|s_andn2_saveexec_b64|s_or_saveexec_b64)\b")


def scan(path):
    hits = []
    kernel = None
    lines = open(path).read().splitlines()
    i = 0
    while i < len(lines):
        m = LABEL.match(lines[i])
        if not m:
            i += 1
            continue
        if m.group(1).startswith("_Z"):
            kernel = m.group(1)
        # the block is entered by falling out of a `s_andn2_b64 exec, exec, sX ; s_cbranch_execnz loop` back edge: exec == 0
        k = i - 1
        while k >= 0 and (not lines[k].strip() or lines[k].lstrip().startswith(";")):
            k -= 1
        loop_exit = k >= 0 and lines[k].strip().startswith("s_cbranch_execnz")
        pending = []
        j = i + 1
        while j < len(lines):
            ln = lines[j]
            if LABEL.match(ln) or ENDS.match(ln):
                break
            if EXEC_OR.match(ln):
                for (k, txt) in pending:
                    hits.append((kernel, m.group(1), k + 1, txt.strip(), ln.strip(), loop_exit))
                break
            if VEC.match(ln):
                pending.append((j, ln))
            j += 1
        i += 1
    return hits


def main():
    bad = 0
    verbose = "-v" in sys.argv
    for p in [x for x in sys.argv[1:] if x != "-v"]:
        hits = scan(p)
        dead = [h for h in hits if h[5]]
        print("%s: %d vector instruction(s) ahead of a block's exec restore; %d of them in a loop-exit block (entered "
              "with exec == 0: they write no lane)" % (p, len(hits), len(dead)))
        for (kernel, label, line, txt, ex, loop_exit) in hits:
            if loop_exit or verbose:
                tag = ("DEAD " if loop_exit else "     ") + ("copy" if COPY.match(txt) else "code")
                print("  %s %s  %s  line %d: %-50s before  %s" % (tag, kernel, label, line, txt, ex))
        bad += len(dead)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
