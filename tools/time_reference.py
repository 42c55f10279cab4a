#!/usr/bin/env python
"""time_reference.py -- BUILD CONTAINER ONLY.  Times the reference's own Node.js path (tools/time_reference.js, the unmodified bundle
under /root/reference) and the C port of it (oracle/) on the SAME core in the same run, for BASELINE configs C1-C5 and the native shape,
and writes profiles/cpu_reference_ratio.json.  bench.py quotes `cpu_baseline.reference_ratio` / `reference_frames_per_s_est` from it."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np          # noqa: E402
import oracle_lib           # noqa: E402
import signals as S         # noqa: E402

CORE = int(os.environ.get("PHAZE_TIMING_CORE", "2"))
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0


def cpu_model():
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            return line.split(":", 1)[1].strip()
    return "unknown"


def time_port(fft, hop, nch, pitch_fn):
    o = oracle_lib.Oracle(fft, hop, nch)
    T = 256
    x = np.stack([S.make_signal("tonal", c, T * hop) for c in range(nch)])
    frames, t0 = 0, time.perf_counter()
    m = 0
    while time.perf_counter() - t0 < SECONDS:
        p = np.array([pitch_fn(m + i) for i in range(T)], np.float32)
        o.process_planar(x, p)
        frames += T * nch
        m += T
    return frames / (time.perf_counter() - t0)


def main():
    os.sched_setaffinity(0, {CORE})
    ref = subprocess.run(["taskset", "-c", str(CORE), "node", os.path.join(ROOT, "tools", "time_reference.js"), str(SECONDS)],
                         capture_output=True, text=True, check=True).stdout.strip().splitlines()
    ref = [json.loads(l) for l in ref]
    pitch = {"C1": lambda m: 1.0, "C2": lambda m: 1.5, "C3": lambda m: 0.8, "C4": lambda m: 1.25,
             "C5": lambda m: 0.5 + 1.5 * (m % 64) / 63, "native": lambda m: 1.5}
    out = {"what": "reference (unmodified www/phase-vocoder.js under Node, one thread) vs the C port oracle/pv_oracle.c, same core, same run",
           "node_version": subprocess.run(["node", "--version"], capture_output=True, text=True).stdout.strip(),
           "cpu_model": cpu_model(), "core": CORE, "seconds_per_measurement": SECONDS, "signal": "tonal (SURVEY section 4)", "configs": []}
    for r in ref:
        fft, hop = (int(v) for v in r["shape"].split("/"))
        port = time_port(fft, hop, r["nch"], pitch[r["config"]])
        out["configs"].append({"config": r["config"], "shape": r["shape"], "nch": r["nch"], "pitch": float(r["pitch"]) if isinstance(r["pitch"], (int, float)) else 0.0,
                               "pitch_desc": r["pitch"], "reference_frames_per_s": r["frames_per_s"], "port_frames_per_s": port,
                               "port_over_reference": port / r["frames_per_s"]})
        print(out["configs"][-1])
    json.dump(out, open(os.path.join(ROOT, "profiles", "cpu_reference_ratio.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
