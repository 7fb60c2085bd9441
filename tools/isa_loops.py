#!/usr/bin/env python3
"""Instruction mix per loop of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).

usage: isa_loops.py listing.s kernel-name-substring
A loop is a backward branch and the label it goes to; every instruction is charged to the innermost
loop around it.  Classes: f64 (v_*_f64), mfma, dpp, mov (v_mov/v_accvgpr), valu (other v_), salu, smem,
lds (ds_), vmem (global/buffer/scratch), scr (scratch_ only), wait, br.
"""
import re
import sys
from collections import Counter


def klass(op, line):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        if "dpp" in line or "quad_perm" in line or "row_" in line:
            return "dpp"
        if op.endswith("_f64") or "_f64_" in op:
            return "f64"
        if op.startswith("v_mov") or op.startswith("v_accvgpr"):
            return "mov"
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "br"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scr"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
        return "vmem"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^\s+(s_cbranch\w+|s_branch)\s+(\.LBB[0-9_]+)", l)
        if m and m.group(2) in labels and labels[m.group(2)] <= i:
            loops.append((labels[m.group(2)], i))
    # merge loops with the same head
    heads = {}
    for a, b in loops:
        heads[a] = max(heads.get(a, 0), b)
    loops = sorted(heads.items())
    def innermost(i):
        best = None
        for a, b in loops:
            if a <= i <= b and (best is None or (b - a) < (best[1] - best[0])):
                best = (a, b)
        return best
    hist = {}
    for i, l in enumerate(body):
        m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
        if not m or l.strip().startswith("."):
            continue
        lp = innermost(i)
        hist.setdefault(lp, Counter())[klass(m.group(1), l)] += 1
    cols = ["f64", "mfma", "dpp", "mov", "valu", "salu", "smem", "lds", "vmem", "scr", "wait", "br"]
    print("%-16s %5s " % ("loop(lines)", "depth") + " ".join("%5s" % c for c in cols))
    for lp in [None] + loops:
        if lp not in hist:
            continue
        depth = 0 if lp is None else sum(1 for a, b in loops if a <= lp[0] and lp[1] <= b)
        name = "outside" if lp is None else "%d-%d" % (lp[0] + start + 1, lp[1] + start + 1)
        print("%-16s %5d " % (name, depth) + " ".join("%5d" % hist[lp][c] for c in cols))


main()
