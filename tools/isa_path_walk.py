#!/usr/bin/env python
"""Walks ONE path through the frame loop of a kernel in a hipcc -S listing and counts the instructions it executes, per phase (the phases are the
s_setprio marks of pv_wave_fft.h's table, in program order).  Branch policy: exec-mask skips (s_cbranch_execz) fall through -- the masked block
runs, as it does when any lane is active --, s_cbranch_execnz is taken, every other conditional branch follows --take / --skip (line numbers
of the extracted function), default: not taken.  Design aid for profiles/r04_instruction_budget.md.
usage: isa_path_walk.py file.s <mangled-name-substring> [--take n,n,...] [--lines]"""
import re
import sys
from collections import Counter, OrderedDict


def extract(lines, key):
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
    end = next((i for i in range(start + 1, len(lines)) if lines[i].startswith("_Z") or lines[i].startswith("\t.section")), len(lines))
    return lines[start:end]


def classify(op):
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("v_"):
        if "_f64" in op:
            return "VALU f64"
        if op.startswith("v_pk_"):
            return "VALU pk"
        if "permlane" in op:
            return "VALU swap"
        if "_dpp" in op or "dpp" in op:
            return "VALU dpp"
        if op.startswith("v_cvt"):
            return "VALU cvt"
        return "VALU other"
    return "other"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    body = extract(lines, sys.argv[2])
    take = set()
    if "--take" in sys.argv:
        take = {int(x) for x in sys.argv[sys.argv.index("--take") + 1].split(",") if x}
    labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
    lh = next(i for i, l in enumerate(body) if "Loop Header" in l)
    head = body[lh].split(":")[0]
    phases = OrderedDict()
    cur = "0 top"
    phases[cur] = Counter()
    i = lh + 1
    steps = 0
    nphase = 0
    dpp_re = re.compile(r"row_|quad_perm|dpp")
    while steps < 20000:
        steps += 1
        t = body[i].strip()
        i += 1
        if not t or t.startswith((";", ".")) and not t.startswith(".LBB"):
            continue
        if t.startswith(".LBB"):
            continue
        tok = t.split()
        op = tok[0]
        if op == "s_setprio":
            nphase += 1
            cur = "%d prio %s (line %d)" % (nphase, tok[1], i)
            phases[cur] = Counter()
        c = classify(op)
        if c.startswith("VALU") and dpp_re.search(t) and "permlane" not in op:
            c = "VALU dpp"
        phases[cur][c] += 1
        if "--lines" in sys.argv:
            print(i, t)
        if op in ("s_branch",) or op.startswith("s_cbranch"):
            tgt = tok[-1]
            taken = op == "s_branch" or op == "s_cbranch_execnz" or (i in take)
            if op == "s_cbranch_execz":
                taken = False
            if taken:
                if tgt == head:
                    break
                i = labels[tgt] + 1
        if op in ("s_endpgm",):
            break
    tot = Counter()
    print("%-28s %6s %6s %6s %6s %6s %6s %6s %6s %6s" % ("phase", "f64", "pk", "swap", "dpp", "cvt", "other", "LDS", "SALU", "VMEM"))
    for k, v in phases.items():
        tot += v
        print("%-28s %6d %6d %6d %6d %6d %6d %6d %6d %6d" % (k, v["VALU f64"], v["VALU pk"], v["VALU swap"], v["VALU dpp"], v["VALU cvt"], v["VALU other"], v["LDS"], v["SALU"], v["VMEM"]))
    v = tot
    print("%-28s %6d %6d %6d %6d %6d %6d %6d %6d %6d" % ("total", v["VALU f64"], v["VALU pk"], v["VALU swap"], v["VALU dpp"], v["VALU cvt"], v["VALU other"], v["LDS"], v["SALU"], v["VMEM"]))
    print("VALU total", sum(c for k, c in v.items() if k.startswith("VALU")))


if __name__ == "__main__":
    main()
