#!/usr/bin/env python
"""pmc_report.py -- turns the rocprofv3 CSVs of profiles/run_profile_r0N.sh into the committed round-2 summaries:
   profiles/<tag>_kernel_trace.md, profiles/<tag>_pmc.md (per-frame counter table of the headline kernel), profiles/<tag>_wg_pmc.md and
   profiles/hbm_traffic.json (FETCH_SIZE x2 + WRITE_SIZE per shape, keyed like bench.py looks them up, stamped with the kernel-source hash).
usage: python profiles/pmc_report.py <gpurun_out/tag> <tag>"""
import csv, re
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    """Identity of the kernel sources: profiles/hbm_traffic.json entries are only reported for the build they were measured on.  Comments and blank lines do not count
    (a reworded comment is not another build; pv_capi.hip, the host runtime, holds no device code and does not count)."""
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "phaze_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) and f != "pv_capi.hip":         # (the host runtime behind the C ABI holds no device code)
            src = open(os.path.join(d, f), "r", encoding="utf-8", errors="replace").read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            code = [re.sub(r"//.*$", "", line).rstrip() for line in src.split("\n")]
            h.update(f.encode() + b"\0" + "\n".join(l for l in code if l.strip()).encode())
    return h.hexdigest()[:16]


def counters(pattern, kernel_sub="pv_"):
    acc = {}
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_sub in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def dominant_counters(pattern):
    """Counters of the kernel that does the work of a launch.  An N = 2048 launch is a pitch scan + pv_wave2k_kernel + a gated pv_wg_kernel, one of the two
    returning at once: take the pv_* kernel (never the scan) with the largest counter total."""
    per = {}
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "pv_" in kn and "pitch_scan" not in kn and "classify" not in kn:
                per.setdefault(kn, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if not per:
        return {}, ""
    name = max(per, key=lambda k: sum(sum(v) for v in per[k].values()))
    return {k: sum(v) / len(v) for k, v in per[name].items()}, name


def kernel_stats(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def main():
    d, tag = sys.argv[1], sys.argv[2]
    # a directory that is not the output of profiles/run_profile_r0N.sh must not overwrite the committed summaries (it happened: a stale scratch directory of the same name)
    need = ["trace", "pmc_1", "hbm_c2_FETCH_SIZE", "hbm_c5_WRITE_SIZE", "hbm_calib_FETCH_SIZE", "wg_c5_1"]
    missing = [n for n in need if not os.path.isdir(os.path.join(d, n))]
    if missing:
        sys.exit(f"{d}: not a run_profile directory (missing {', '.join(missing)}); nothing written")
    out = []
    # ---- kernel trace ----
    st = kernel_stats(os.path.join(d, "trace"))
    lines = [f"# {tag}: rocprofv3 --kernel-trace --stats of `python bench.py --no-extras --no-cpu-baseline --steps 40 --warmup 10`", "",
             "| kernel | calls | total ms | avg us | % | min us | max us |", "|---|---|---|---|---|---|---|"]
    for r in st:
        lines.append(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} | "
                     f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} |")
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "pv_" in r["Kernel_Name"] and "classify" not in r["Kernel_Name"]]
        # round 4: an N = 1024 batch launch is a classification kernel + two instances of pv_wave_kernel_1024, one of which returns at once:
        # the statistics below are those of the instance that does the work (the dispatches of the kernel name with the largest total time)
        tot = {}
        for r in rows:
            tot[r["Kernel_Name"]] = tot.get(r["Kernel_Name"], 0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if tot:
            kbest = max(tot, key=tot.get)
            rows = [r for r in rows if r["Kernel_Name"] == kbest]
            lines += ["", f"dominant kernel: `{kbest[:110]}`"]
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
        if durs:
            steady = durs[len(durs) // 3:]
            r = rows[-1]
            lines += ["", f"pv_* dispatches: {len(durs)}; mean {sum(durs)/len(durs):.2f} us, mean of the last two thirds (clock settled) {sum(steady)/len(steady):.2f} us; "
                          f"grid {r['Grid_Size_X']}, workgroup {r['Workgroup_Size_X']}, LDS {r['LDS_Block_Size']} B, VGPR {r['VGPR_Count']}, scratch {r['Scratch_Size']} B/lane"]
    open(os.path.join(ROOT, "profiles", f"{tag}_kernel_trace.md"), "w").write("\n".join(lines) + "\n")
    # ---- headline PMC ----
    c, _kn = dominant_counters(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"))
    frames = None
    try:
        j = json.loads([l for l in open(os.path.join(d, "pmc_1.log")) if l.startswith("{")][-1])
        fpc, hops = j["config"]["frames_per_chunk"], j["config"]["hops_per_step"]
        chains = -(-hops // fpc)
        frames = hops + (chains - 1) * 3                       # R - 1 = 3 halo frames per chain but the first
    except Exception:
        pass
    lines = [f"# {tag}: PMC passes of the headline kernel (separate rocprofv3 --pmc runs, profiles/run_profile_r0N.sh)", "",
             f"computed frames per launch: {frames}", "", "| counter | mean per dispatch | per computed frame |", "|---|---|---|"]
    for k in sorted(c):
        lines.append(f"| {k} | {c[k]:.6g} | {c[k]/frames:.1f} |" if frames else f"| {k} | {c[k]:.6g} | |")
    open(os.path.join(ROOT, "profiles", f"{tag}_pmc_table.md"), "w").write("\n".join(lines) + "\n")
    # ---- HBM traffic ----
    tj_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    tj = {}
    sha = csrc_sha16()
    calib = {}
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        cc, _ = counters(os.path.join(d, f"hbm_calib_{cn}", "**", "*counter_collection.csv"), "")
        # the copy kernel is the largest dispatch: take the max over kernels instead
        best = 0.0
        for f in glob.glob(os.path.join(d, f"hbm_calib_{cn}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == cn:
                    best = max(best, float(r["Counter_Value"]))
        calib[cn] = best
    shapes = {"c2": (1024, 256, 1, 1 << 20), "c3": (2048, 512, 2, 262144), "c4": (4096, 1024, 1024, 64), "c5": (8192, 2048, 8, 16384), "native": (2048, 128, 2, 262144)}
    for name, (fft, hop, nch, hops) in shapes.items():
        fs, kname = dominant_counters(os.path.join(d, f"hbm_{name}_FETCH_SIZE", "**", "*counter_collection.csv"))
        ws, _ = dominant_counters(os.path.join(d, f"hbm_{name}_WRITE_SIZE", "**", "*counter_collection.csv"))
        if "FETCH_SIZE" not in fs or "WRITE_SIZE" not in ws:
            continue
        fetch = fs["FETCH_SIZE"] * 1024 * 2                      # KiB, gfx950 wide-read correction (MI355X_MICROARCH.md, HBM section)
        write = ws["WRITE_SIZE"] * 1024
        alg = nch * hops * 2 * hop * 4
        tj[f"{fft}/{hop}/ch{nch}/hops{hops}"] = {"bytes_per_launch": int(fetch + write), "fetch_bytes_corrected_x2": int(fetch), "write_bytes": int(write),
                                                  "algorithmic_bytes": alg, "traffic_over_algorithmic": (fetch + write) / alg, "kernel": (re.search(r"pv_\w+", kname) or [""])[0], "csrc_sha16": sha,
                                                  "source": f"profiles/run_profile_r0N.sh {tag}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE (KiB) doubled per the gfx950 wide-read rule"}
    tj["_calibration"] = {"what": "torch copy_ of 1 GiB (4 dispatches) under the same counters: largest dispatch", "FETCH_SIZE_KiB": calib.get("FETCH_SIZE"),
                          "WRITE_SIZE_KiB": calib.get("WRITE_SIZE"), "expected_KiB": 1 << 20,
                          "fetch_x2_over_expected": (calib.get("FETCH_SIZE", 0) * 2) / (1 << 20), "write_over_expected": calib.get("WRITE_SIZE", 0) / (1 << 20)}
    json.dump(tj, open(tj_path, "w"), indent=1)
    # ---- what binds: VALU issue cycles over the SIMD cycles of the launch (bench.py: roofline_valu), per shape, stamped like the traffic ----
    vj = {}
    def valu_entry(c, frames):
        if not c or "SQ_ACTIVE_INST_VALU" not in c or "GRBM_GUI_ACTIVE" not in c:
            return None
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0                         # summed over the 8 XCDs
        e = {"frac": c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc), "active_inst_valu_quad_cycles": c["SQ_ACTIVE_INST_VALU"], "launch_cycles": cyc, "simds": 1024,
             "valu_insts_per_frame": (c.get("SQ_INSTS_VALU", 0.0) / frames) if frames else None,
             "wait_any_frac": (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]) if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c else None, "csrc_sha16": sha,
             "source": f"profiles/run_profile_r0N.sh {tag}: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), separate --pmc passes"}
        return e
    hc, _ = dominant_counters(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"))
    e = valu_entry(hc, frames)
    if e:
        vj["1024/256/ch1/hops1048576"] = e
    wgshapes = {"c3": (2048, 512, 2, 262144), "c4": (4096, 1024, 1024, 64), "c5": (8192, 2048, 8, 16384), "native": (2048, 128, 2, 262144)}
    for name, (fft, hop, nch, hops) in wgshapes.items():
        cc, _k = dominant_counters(os.path.join(d, f"wg_{name}_*", "**", "*counter_collection.csv"))
        fr = None
        try:
            j = json.loads([l for l in open(os.path.join(d, f"wg_{name}_1.log")) if l.startswith("{")][-1])
            cfg = j["config"]
            chains = -(-cfg["hops_per_step"] // cfg["frames_per_chunk"])
            fr = cfg["channels"] * (cfg["hops_per_step"] + (chains - 1) * (cfg["fft"] // cfg["hop"] - 1))
        except Exception:
            pass
        e = valu_entry(cc, fr)
        if e:
            vj[f"{fft}/{hop}/ch{nch}/hops{hops}"] = e
    if vj:
        json.dump(vj, open(os.path.join(ROOT, "profiles", "valu_issue.json"), "w"), indent=1)
    # ---- workgroup kernel ----
    lines = [f"# {tag}: counters of the other shapes' kernels (pv_wave2k_kernel at N = 2048, pv_wg_kernel above), per computed frame (separate --pmc passes)", ""]
    for name in ("c3", "c3f15", "c3f07", "c4", "c5", "c5f08", "c5sweep", "native", "c2f08"):
        c, kname = dominant_counters(os.path.join(d, f"wg_{name}_*", "**", "*counter_collection.csv"))
        if not c:
            continue
        try:
            j = json.loads([l for l in open(os.path.join(d, f"wg_{name}_1.log")) if l.startswith("{")][-1])
            cfg = j["config"]
            R = cfg["fft"] // cfg["hop"]
            chains = -(-cfg["hops_per_step"] // cfg["frames_per_chunk"])
            frames = cfg["channels"] * (cfg["hops_per_step"] + (chains - 1) * (R - 1))
            head = f"## {name} (`{(re.search(r'pv_[A-Za-z0-9_]+', kname) or [''])[0]}`): {cfg['workload'][:110]} -- {j['roofline']['kernel_ms']:.3f} ms, {j['value']:.4g} frames/s, {100*j['roofline']['frac']:.2f} % of 8 TB/s"
        except Exception:
            frames, head = None, f"## {name}"
        lines += [head, "", "| counter | per dispatch | per computed frame |", "|---|---|---|"]
        for k in sorted(c):
            lines.append(f"| {k} | {c[k]:.5g} | {c[k]/frames:.1f} |" if frames else f"| {k} | {c[k]:.5g} | |")
        lines.append("")
    open(os.path.join(ROOT, "profiles", f"{tag}_wg_pmc.md"), "w").write("\n".join(lines) + "\n")
    print("wrote profiles for", tag, "csrc", sha)


if __name__ == "__main__":
    main()
