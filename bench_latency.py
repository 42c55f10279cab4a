#!/usr/bin/env python
"""bench_latency.py -- per-call latency histogram of the STREAMING form (pv_process: one render quantum per call).

SURVEY 8(f)-1 / BASELINE configs[4]: 96 kHz 8-ch, FFT=8192 hop=2048, pitchFactor swept 0.5 -> 2.0, per-frame latency
histogram.  Each call = copy-in (host blocks -> pinned) + H2D + kernel + D2H + copy-out, synchronous, timed on the host
around the C-ABI call (ctypes adds ~2 us).  Prints one JSON line per configuration.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(fft, hop, nch, fs, calls, sweep, flags=0):
    import numpy as np
    import phaze_amd
    import signals as S
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1, flags=flags)
    L = pv._L
    x = np.stack([S.make_signal("tonal", c, 64 * hop) for c in range(nch)])
    fpt = C.POINTER(C.c_float)
    outs = [np.zeros(hop, np.float32) for _ in range(nch)]
    op = (fpt * nch)(*[o.ctypes.data_as(fpt) for o in outs])
    blocks = [[np.ascontiguousarray(x[c, (m % 64) * hop:((m % 64) + 1) * hop]) for c in range(nch)] for m in range(64)]
    ips = [(fpt * nch)(*[b.ctypes.data_as(fpt) for b in blocks[m]]) for m in range(64)]
    lat = np.empty(calls, np.float64)
    for m in range(calls + 50):
        pf = (0.5 + 1.5 * ((m % 256) / 255.0)) if sweep else 1.5
        t0 = time.perf_counter_ns()
        rc = L.pv_process(pv._h, ips[m % 64], op, nch, hop, C.c_float(pf))
        t1 = time.perf_counter_ns()
        assert rc == 0
        if m >= 50:
            lat[m - 50] = (t1 - t0) * 1e-3
    pv.close()
    q = lambda p: float(np.percentile(lat, p))
    edges = [0, 25, 50, 75, 100, 150, 200, 300, 500, 1000, 1e9]
    hist = np.histogram(lat, bins=edges)[0].tolist()
    budget_us = hop / fs * 1e6
    return {"metric": "stream_call_latency_us", "config": {"workload": f"{nch}-ch {fs // 1000} kHz FFT={fft} hop={hop} " + ("pitchFactor sweep 0.5->2.0" if sweep else "pitchFactor 1.5"),
                                                            "calls": calls, "wait": "stream synchronize (PV_FLAG_STREAM_EVENT_WAIT)" if flags & 8 else "resident kernel (PV_FLAG_PERSISTENT_STREAM)" if flags & 32 else "completion words in pinned memory (default)",
                                                            "input": "kernel reads pinned host memory (PV_FLAG_STREAM_PINNED_INPUT)" if flags & 16 else
                                                                     "host writes device memory through the BAR when the quantum is <= 16 KB (default on a large-BAR device)"},
            "p50": q(50), "p90": q(90), "p99": q(99), "max": float(lat.max()), "mean": float(lat.mean()),
            "realtime_budget_us": budget_us, "budget_over_p99": budget_us / q(99),
            "histogram_us_edges": edges[:-1] + ["inf"], "histogram_counts": hist,
            "frames_per_s_streaming": nch / (lat.mean() * 1e-6)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=3000)
    ap.add_argument("--event-wait", action="store_true", help="also run every configuration with PV_FLAG_STREAM_EVENT_WAIT (the round-2 wait) for the A/B")
    ap.add_argument("--pinned-input", action="store_true", help="also run every configuration with PV_FLAG_STREAM_PINNED_INPUT (kernel reads the hop over PCIe) for the A/B")
    ap.add_argument("--resident", action="store_true", help="also run every configuration with PV_FLAG_PERSISTENT_STREAM (resident kernel where the shape supports it)")
    args = ap.parse_args()
    for cfg in [(8192, 2048, 8, 96000, True), (2048, 128, 2, 48000, False), (1024, 256, 1, 48000, False), (4096, 1024, 8, 48000, False)]:
        print(json.dumps(run(*cfg[:4], args.calls, cfg[4])), flush=True)
        if args.event_wait:
            print(json.dumps(run(*cfg[:4], args.calls, cfg[4], flags=8)), flush=True)
        if args.pinned_input:
            print(json.dumps(run(*cfg[:4], args.calls, cfg[4], flags=16)), flush=True)
        if args.resident:
            print(json.dumps(run(*cfg[:4], args.calls, cfg[4], flags=32)), flush=True)


if __name__ == "__main__":
    main()
