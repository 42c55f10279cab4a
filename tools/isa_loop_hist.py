#!/usr/bin/env python
"""Static instruction histogram of the frame loop of a kernel in a hipcc -S listing.
usage: isa_loop_hist.py file.s <mangled-name-substring> [--lines]   (design aid)"""
import sys
from collections import Counter


def function_body(lines, key):
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i] or "s_setpc_b64" in lines[i])
    return start, end


def classify(op):
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_"):
        return "SALU/ctl"
    if op.startswith("v_"):
        if "_f64" in op:
            return "VALU f64"
        if op.startswith("v_pk_"):
            return "VALU packed"
        if "readlane" in op or "writelane" in op or "readfirstlane" in op:
            return "VALU lane"
        return "VALU other"
    return "other"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    start, end = function_body(lines, sys.argv[2])
    body = lines[start:end]
    lh = next(i for i, l in enumerate(body) if "Loop Header" in l)
    # the loop ends at the backward branch to its header label
    label = body[lh].split(":")[0].strip()
    le = max(i for i, l in enumerate(body) if l.strip().startswith(("s_cbranch", "s_branch")) and label in l.split()[-1] and i > lh)
    ops, cls = Counter(), Counter()
    for l in body[lh:le + 1]:
        t = l.strip().split()
        if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
            continue
        ops[t[0]] += 1
        cls[classify(t[0])] += 1
    print(f"{sys.argv[2]}: loop {label} lines {start + lh + 1}..{start + le + 1}, {sum(ops.values())} static instructions")
    for k, v in sorted(cls.items(), key=lambda x: -x[1]):
        print(f"  {v:5d}  {k}")
    if "--ops" in sys.argv:
        for k, v in sorted(ops.items(), key=lambda x: -x[1]):
            print(f"    {v:5d}  {k}")


if __name__ == "__main__":
    main()
