#!/usr/bin/env python
"""Condense a rocprofv3 output directory (CSV format) into a short, committable summary.

usage: python profiles/summarize.py <rocprof_dir> <out.md> [title]
Keeps: per-kernel stats (names truncated), the pv_* dispatch rows of the kernel trace (grid, LDS, VGPR),
and, when present, per-dispatch PMC counter values of the pv_* kernels.
"""
import csv
import glob
import os
import sys


def short(name, n=90):
    return name if len(name) <= n else name[:n] + "..."


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(d)
    lines = [f"# {title}", ""]
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        lines += ["## kernel stats (rocprofv3 --kernel-trace --stats)", "", "| kernel | calls | total ms | avg us | % | min us | max us |", "|---|---|---|---|---|---|---|"]
        for r in csv.DictReader(open(f)):
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | "
                         f"{float(r['Percentage']):.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} |")
        lines.append("")
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if "pv_" in r["Kernel_Name"]]
        if rows:
            lines += ["## pv_* dispatches (kernel trace)", "", "| kernel | dur us | grid | wg | LDS B | VGPR | SGPR | scratch |", "|---|---|---|---|---|---|---|---|"]
            for r in rows[:12]:
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                lines.append(f"| `{short(r['Kernel_Name'], 60)}` | {dur:.2f} | {r['Grid_Size_X']}x{r['Grid_Size_Y']} | {r['Workgroup_Size_X']} | "
                             f"{r['LDS_Block_Size']} | {r['VGPR_Count']} | {r['SGPR_Count']} | {r['Scratch_Size']} |")
            lines.append("")
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "pv_" not in r["Kernel_Name"]:
                continue
            key = (short(r["Kernel_Name"], 60), r["Counter_Name"])
            acc.setdefault(key, []).append(float(r["Counter_Value"]))
        if acc:
            lines += [f"## PMC counters ({os.path.basename(f)})", "", "| kernel | counter | dispatches | mean per dispatch |", "|---|---|---|---|"]
            for (k, c), v in sorted(acc.items()):
                lines.append(f"| `{k}` | {c} | {len(v)} | {sum(v)/len(v):.6g} |")
            lines.append("")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
