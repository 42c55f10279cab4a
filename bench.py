#!/usr/bin/env python
"""bench.py -- headline measurement of the phaze hot path on MI355X.

A "step" = one pass of the hot path (one pv_process_batch_device launch) over one batch of synthetic input that is already
resident in HBM.  The default workload is BASELINE.json configs[1]: mono 48 kHz, FFT=1024, hop=256, pitchFactor=1.5, run in
throughput mode: one long stream (HOPS hops per step) is processed frame-parallel (chunks of frames with an (R-1)-frame halo, see
DESIGN.md).

    python bench.py --gpus N --steps K --warmup W

N > 1: every rank owns an independent stream of the same size on its own GPU (weak scaling, no data-path collective; RCCL only
carries the barrier and the max-over-ranks reduction of the timing).  Under `torch.distributed.run` the ranks come from the
environment; a plain `python bench.py --gpus N` spawns the N ranks itself (one process per GPU).  On a box with fewer than N GPUs
only the available devices are measured and the line says so (`requested_gpus`, `replicas_measured`).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     algorithmic HBM bytes per launch / measured launch duration vs the 8 TB/s peak (HIP events on the launch stream)
  "cpu_baseline": the CPU oracle (a C port of the reference JS, kind "port") timed on ONE host core on a bounded sample of the same
                  workload, plus the estimate for the reference's own Node.js path (ratio measured by tools/time_reference.js)
  "configs":      one short measured line per other BASELINE config (C3, a C4 share, C5 with its pitch sweep, the 8-channel form of the
                  headline shape) and "latency_us": the streaming-quantum histogram of C5 (one launch per quantum; "latency_us_resident": the same on the resident kernel), "latency_us_headline_shape": mono 1024/256 per quantum (launch form / resident kernel) -- N = 1 only, skipped with --no-extras
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
DTYPE = "f32+f64"           # forward FFT in f32 first, re-run in f64 for the frames where a peak decision is within the f32 transform's error (N = 1024 and N = 2048:
                            # pv_wave_kernel_1024, pv_wave2k_kernel; N >= 4096: f64 forward); shift / inverse FFT / overlap-add in f32, I/O f32
LINE_BUDGET = 7000          # bytes: the driver keeps the last 8 KB of stdout; the ONE line must fit with room to spare (round-5 verdict, item 5).  What the short ids
                            # ("C3", "C4-share", ...) stand for is spelled out in profiles/bench_workloads.md; the unabridged record of a run goes to bench_detail.json


def synth_input(torch, nch, nsamples, device, seed):
    """Tonal partials + a -36 dB noise floor (always keep a noise floor, K12), generated on device (a handful of launches
    whatever the channel count: the profiles stay small)."""
    g = torch.Generator(device=device)
    g.manual_seed(1234 + seed)
    base = 2 * 3.14159265358979 / 48000.0
    c = torch.arange(nch, device=device, dtype=torch.float32)[:, None]
    x = torch.empty((nch, nsamples), device=device, dtype=torch.float32)
    blk = max(1, (1 << 26) // max(nsamples, 1))                    # channels per block: bounds the temporaries to ~256 MB each
    for c0 in range(0, nch, blk):
        cc = c[c0:c0 + blk]
        n = torch.arange(nsamples, device=device, dtype=torch.float32)[None, :]
        s = 0.25 * torch.sin(n * (base * (220.0 + 17 * (cc % 97)))) + 0.125 * torch.sin(n * (base * (1375.0 + 5 * (cc % 89)))) \
            + 0.0625 * torch.sin(n * (base * 6857.0))
        s += (torch.rand(s.shape, device=device, generator=g) - 0.5) * (2.0 / 64)
        x[c0:c0 + blk] = s
    return x


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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(fft, hop, pitch, x_prefix, what, target_seconds=12.0):
    """Times the CPU oracle (oracle/pv_oracle.c -- the checker, used here only as the reported baseline) on one core, on a prefix of the SAME
    signal the GPU leg processed (x_prefix: channel 0 of the headline input, downloaded from HBM)."""
    import numpy as np
    import oracle_lib
    avail = x_prefix.shape[-1] // hop
    probe = min(1024, avail)
    x = np.ascontiguousarray(x_prefix[:1, :probe * hop])
    p = np.full(probe, pitch, np.float32)
    o = oracle_lib.Oracle(fft, hop, 1)
    t0 = time.perf_counter()
    o.process_planar(x, p)
    dt = time.perf_counter() - t0
    n = int(max(probe, min(avail, target_seconds / (dt / probe))))
    x = np.ascontiguousarray(x_prefix[:1, :n * hop])
    p = np.full(n, pitch, np.float32)
    o = oracle_lib.Oracle(fft, hop, 1)
    t0 = time.perf_counter()
    o.process_planar(x, p)
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"the first {n} hops of channel 0 of {what} (downloaded from HBM), mono {fft}/{hop} pf={pitch}, single thread, {dt:.1f} s",
           "host_cpus": os.cpu_count(), "cpu_model": cpu_model()}
    # the reference itself (unmodified JS bundle under Node, one thread) cannot travel to the GPU box; tools/time_reference.js timed it
    # next to this port on the same core of the build container -> port/reference ratio -> estimate for this host
    rpath = os.path.join(ROOT, "profiles", "cpu_reference_ratio.json")
    if os.path.exists(rpath):
        try:
            rj = json.load(open(rpath))
            key = f"{fft}/{hop}"
            ent = next((c for c in rj["configs"] if c["shape"] == key and abs(c["pitch"] - pitch) < 1e-6), None) or \
                next((c for c in rj["configs"] if c["shape"] == key), None)
            if ent:
                out["reference_ratio"] = ent["port_over_reference"]
                out["reference_frames_per_s_est"] = out["value"] / ent["port_over_reference"]
                out["reference_ratio_source"] = f"profiles/cpu_reference_ratio.json ({rj.get('node_version', 'node')}, {rj.get('cpu_model', '?')}; tools/time_reference.js)"
        except Exception:
            pass
    return out


def other_input(torch, kind, nch, nsamples, device, seed):
    """The signal classes next to synth_input that bound the fp32-first forward transform: white noise (nothing falls back), two clean partials over a -80 / -60 dB
    floor (every / most frames fall back), 16-bit quantised material ("q16": two partials + dither rounded to 1/32768) and digital silence -- tools/flip_count.py's classes."""
    g = torch.Generator(device=device)
    g.manual_seed(4321 + seed)
    pi2 = 2 * 3.14159265358979
    if kind == "white":
        return (torch.rand((nch, nsamples), device=device, generator=g) - 0.5).float()
    if kind == "silence":
        return torch.zeros((nch, nsamples), device=device, dtype=torch.float32)
    i = torch.arange(nsamples, device=device, dtype=torch.float64)[None, :]
    if kind == "q16":
        x = (0.4 * torch.sin(pi2 * i * 0.031) + 0.2 * torch.sin(pi2 * i * 0.177)).float().expand(nch, nsamples).clone()
        x += (torch.rand((nch, nsamples), device=device, generator=g) - 0.5) / 32768
        return torch.round(x * 32768) / 32768
    if kind.startswith("tonal"):
        amp = 10.0 ** (-float(kind[5:]) / 20.0)
        x = (0.5 * torch.sin(pi2 * i * 0.0123) + 0.3 * torch.sin(pi2 * i * 0.0931)).float().expand(nch, nsamples).clone()
        x += (torch.rand((nch, nsamples), device=device, generator=g) * 2 - 1) * amp
        return x
    raise ValueError(kind)


def measure(torch, phaze_amd, dev, dist, fft, hop, nch, T, pitch_t, steps, warmup, label, local_rank, frames_per_chunk=0,
            pitch_stride=0, ch_per_stream=1, parity_hops=0, seed=0, repeats=1, keep_prefix_hops=0, flags=0, signal="bench", preheat_s=0.0):
    """One workload: resident input, `warmup` + `steps` launches bracketed by HIP events on the launch stream.  Returns a dict."""
    import numpy as np
    from phaze_amd import shard
    x = synth_input(torch, nch, T * hop, dev, seed) if signal == "bench" else other_input(torch, signal, nch, T * hop, dev, seed)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1, device_id=local_rank, frames_per_chunk=frames_per_chunk, flags=flags)
    stream = torch.cuda.Stream(device=dev)        # a real (non-null) torch stream: the library launches on it, so HIP events bracket the kernels
    assert stream.cuda_stream != 0
    pv.set_stream(stream.cuda_stream)

    def step():
        pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, pitch_t.data_ptr(), pitch_stride, ch_per_stream)

    parity = None
    if parity_hops:
        import oracle_lib
        K = min(parity_hops, T)
        pv.reset()
        step()
        pv.synchronize()
        got = y[:, :K * hop].cpu().numpy().astype(np.float64)
        ph = pitch_t.cpu().numpy()
        nstreams = nch // ch_per_stream
        err2, cnt = 0.0, 0
        for s in range(nstreams):          # one oracle instance per stream (streams are independent processors with their own pitch row)
            rows = slice(s * ch_per_stream, (s + 1) * ch_per_stream)
            prow = ph[s * pitch_stride:s * pitch_stride + K] if pitch_stride else ph[:K]
            ref = oracle_lib.Oracle(fft, hop, ch_per_stream).process_planar(x[rows, :K * hop].cpu().numpy(), np.ascontiguousarray(prow))
            err2 += float(np.sum((got[rows] - ref) ** 2)); cnt += ref.size
            if s >= 3:
                break                       # a few streams are enough for a spot check
        parity = float(np.sqrt(err2 / cnt))
    pv.reset()
    pv.forward_stats(reset=True)
    regions = []
    with torch.cuda.stream(stream):
        if preheat_s > 0:                         # (see --preheat-seconds: the same workload, untimed, until the box has been busy that long)
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < preheat_s:
                for _ in range(16):
                    step()
                torch.cuda.synchronize()
        for _ in range(warmup):
            step()
        for _ in range(max(1, repeats)):          # every region: EXACTLY `steps` launches between barrier + synchronize on both sides
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record(stream)
            for _ in range(steps):
                step()
            ev1.record(stream)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            km = ev0.elapsed_time(ev1) / steps                # HIP events on the launch stream: average launch duration
            regions.append(tuple(shard.reduce_max([el, km], dist, dev)))      # MAX over ranks
    # the line reports the MEDIAN region (box noise is 1-2 % between regions, 3-4 % between boxes); all regions, their min and max are listed next to it
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    elapsed, kernel_ms = regions[order[len(order) // 2]]
    info = pv.info()
    fwd_frames, fwd_fallbacks = pv.forward_stats()            # frames whose forward transform ran fp32-first in the launches above, and how many of them re-ran it in fp64
    pv.close()
    frames = nch * T
    alg_bytes = frames * 2 * hop * 4                          # 2*hop*4 B per channel-frame
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
    prefix = x[:1, :min(keep_prefix_hops, T) * hop].cpu().numpy() if keep_prefix_hops else None
    del x, y
    return {"label": label, "frames_per_step_rank": frames, "elapsed": elapsed, "kernel_ms": kernel_ms, "info": info,
            "alg_bytes": alg_bytes, "achieved_gbs": achieved, "parity": parity, "x_prefix": prefix,
            "fallback_rate": (fwd_fallbacks / fwd_frames) if fwd_frames else None, "fp32_first_frames": fwd_frames,
            "regions_ms_per_step": [r[0] / steps * 1e3 for r in regions], "regions_kernel_ms": [r[1] for r in regions]}


def pcie_bandwidth(torch, dev, nbytes=256 << 20, reps=4):
    """Pinned hipMemcpy bandwidth of this box, GB/s: host->device and device->host alone, and each way with BOTH directions in flight (what a
    pipelined host-buffer batch can reach at best).  Measured in the same run as the host-buffer lines that are priced against it."""
    n = nbytes // 4
    ha, hb = torch.empty(n, dtype=torch.float32).pin_memory(), torch.empty(n, dtype=torch.float32).pin_memory()
    da, db = torch.empty(n, dtype=torch.float32, device=dev), torch.zeros(n, dtype=torch.float32, device=dev)
    ha.fill_(1.0)
    hb.zero_()                                                            # touched once: the first device-to-host copy must not pay for the pages
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def run(up, down):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if up:
                with torch.cuda.stream(s1):
                    da.copy_(ha, non_blocking=True)
            if down:
                with torch.cuda.stream(s2):
                    hb.copy_(db, non_blocking=True)
        torch.cuda.synchronize()
        return reps * nbytes / (time.perf_counter() - t0) / 1e9
    run(True, True)
    best = lambda up, down: max(run(up, down) for _ in range(3))          # the link's capability: best of three rounds of `reps` copies
    bw = {"h2d_alone": best(True, False), "d2h_alone": best(False, True), "each_way_concurrent": best(True, True), "bytes": nbytes, "rounds": 3}
    # What a direction of the link can carry: the best of the three figures.  The probe's own numbers move from box to box (torch's two copy streams reached
    # 49 GB/s each way on one box and 29 on the next, whose single directions ran at 57): priced against the concurrent figure alone the batch came out at 153 %.
    bw["reference"] = max(bw["h2d_alone"], bw["d2h_alone"], bw["each_way_concurrent"])
    return bw


def host_batch(torch, phaze_amd, dev, fft, hop, nch, T, cps, pitch_rows, steps, local_rank, bw, workload):
    """The PRODUCT boundary with host pointers (what a Node / C host calls): pv_process_batch on buffers in page-locked memory (pv_host_alloc),
    wall clock around `steps` synchronous calls, PCIe both ways included.  Checked bit for bit against the HBM-resident form of the same batch."""
    import numpy as np
    n = T * hop
    xd = synth_input(torch, nch, n, dev, seed=3)
    x, y = phaze_amd.pinned_empty((nch, n)), phaze_amd.pinned_empty((nch, n))
    x[:] = xd.cpu().numpy()
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T, device_id=local_rank)
    pv.process_batch(x, pitch_rows, channels_per_stream=cps, out=y)              # warm-up: streams, events, first launches
    pv.reset()
    pv.process_batch(x, pitch_rows, channels_per_stream=cps, out=y)
    first = y.copy()
    t0 = time.perf_counter()
    for _ in range(steps):
        pv.process_batch(x, pitch_rows, channels_per_stream=cps, out=y)
    dt = time.perf_counter() - t0
    pv.close()
    # the same batch, resident: one launch on device buffers from a fresh handle
    ref = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1, device_id=local_rank)
    yd = torch.empty_like(xd)
    pd = torch.from_numpy(np.ascontiguousarray(pitch_rows)).to(dev)
    ref.process_batch_device(xd.data_ptr(), yd.data_ptr(), nch, T, n, pd.data_ptr(), pitch_rows.shape[1] if pitch_rows.ndim == 2 else 0, cps)
    ref.synchronize()
    ref.close()
    same = bool(np.array_equal(first.view(np.uint32), yd.cpu().numpy().view(np.uint32)))
    gbs = steps * nch * n * 4 / dt / 1e9
    return {"workload": workload, "value": steps * nch * T / dt, "unit": "frames/s", "steps": steps, "ms_per_step": dt / steps * 1e3,
            "form": "pv_process_batch on host pointers in page-locked memory (pv_host_alloc), synchronous calls, wall clock: PCIe both ways + kernels, pipelined inside the call",
            "gbytes_per_s_each_way": gbs, "pcie_frac": gbs / bw["reference"], "pcie_pinned_memcpy_gbs": bw,
            "bit_equal_to_resident_form": same}


def node_sharded_line(bw, streams, cps, fft, hop, T, steps):
    """The same share through the Node.js boundary: tools/bench_node_sharded.js (ShardedPhaseVocoder.processInPlace -> N-API -> pv_process_batch)."""
    import shutil
    node = shutil.which("node")
    if not node or not os.path.exists(os.path.join(ROOT, "phaze_amd", "node", "phaze_napi.node")):
        return None
    r = subprocess.run([node, os.path.join(ROOT, "tools", "bench_node_sharded.js"), "--gpus", "1", "--streams", str(streams), "--channels", str(cps), "--fft", str(fft),
                        "--hop", str(hop), "--hops", str(T), "--steps", str(steps)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    if r.returncode != 0:
        return {"workload": "C4 share through sharded.js", "error": (r.stderr or r.stdout)[-400:]}
    j = json.loads(r.stdout.strip().splitlines()[-1])
    return {"workload": f"BASELINE configs[3], one GPU's share through the Node.js boundary (phaze_amd/node/sharded.js): {j['config']['workload']}",
            "value": j["value"], "unit": "frames/s", "steps": j["steps"], "ms_per_step": j["ms_per_step"], "form": j["form"],
            "gbytes_per_s_each_way": j["gbytes_per_s_each_way"], "pcie_frac": j["gbytes_per_s_each_way"] / bw["reference"], "node": j["node"],
            "shards_in_flight_together": j["shards_in_flight_together"]}


def synth_stream(torch, nch, n0, n1, device):
    """Samples [n0, n1) of ONE deterministic stream (a function of the absolute sample index only), so that every rank of a
    time-sharded run cuts its span out of the same signal: the partials of synth_input + an index-hashed noise floor."""
    n = torch.arange(n0, n1, device=device, dtype=torch.float64)
    base = 2 * 3.14159265358979 / 48000.0
    x = torch.empty((nch, n1 - n0), device=device, dtype=torch.float32)
    for c in range(nch):
        s = 0.25 * torch.sin(n * (base * (220.0 + 17 * c))) + 0.125 * torch.sin(n * (base * (1375.0 + 5 * c))) + 0.0625 * torch.sin(n * (base * 6857.0))
        h = torch.frac(torch.sin(n * 12.9898 + 78.233 * (c + 1)) * 43758.5453)
        x[c] = (s + (h - 0.5) * (2.0 / 64)).to(torch.float32)
    return x


def measure_time_shard(torch, phaze_amd, dev, dist, fft, hop, nch, T, pitch_value, steps, warmup, local_rank, rank, world):
    """ONE stream of T hops split along the time axis over the ranks: rank r owns hops [r T / W, (r + 1) T / W), imports
    {input tail, acc = 0, timeCursor} R - 1 hops before its span (pv_import_state) and recomputes that halo -- no hand-over between ranks.
    Parity: the first hops of every span against the oracle fed the same stream prefix is too slow for spans deep in the stream, so each rank
    checks its span start against a handle that ran a longer lead-in (state reached by processing, not by import)."""
    import numpy as np
    from phaze_amd import shard
    R, L = fft // hop, fft - hop
    lo, hi = rank * T // world, (rank + 1) * T // world
    start = max(lo - (R - 1), 0)
    x = synth_stream(torch, nch, start * hop, hi * hop, dev)
    y = torch.empty_like(x)
    Ts = hi - start
    pitch = torch.full((Ts,), pitch_value, device=dev, dtype=torch.float32)
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1, device_id=local_rank)
    stream = torch.cuda.Stream(device=dev)
    pv.set_stream(stream.cuda_stream)

    def import_span_state():
        pv.reset()
        if start > 0:
            tail = synth_stream(torch, nch, start * hop - L, start * hop, dev).cpu().numpy()
            for c in range(nch):
                pv.import_state(c, hist=tail[c], acc=np.zeros(L, np.float32), time_cursor=start * hop)

    def step():
        pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, Ts, Ts * hop, pitch.data_ptr(), 0, 1)

    # parity of the hand-over-free start: a second handle reaches hop `lo` by PROCESSING a lead-in of 64 hops (no import of the
    # accumulator), and must produce the same first hops of the span bit for bit
    import_span_state()
    step()
    pv.synchronize()
    K = min(32, hi - lo)
    got = y[:, (lo - start) * hop:(lo - start + K) * hop].cpu().numpy()
    lead = min(lo, 64 + R)
    chk = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=lead + K, device_id=local_rank)
    if lo - lead > 0:
        tail = synth_stream(torch, nch, (lo - lead) * hop - L, (lo - lead) * hop, dev).cpu().numpy()
        for c in range(nch):
            chk.import_state(c, hist=tail[c], acc=np.zeros(L, np.float32), time_cursor=(lo - lead) * hop)
    xs = synth_stream(torch, nch, (lo - lead) * hop, (lo + K) * hop, dev).cpu().numpy()
    ref = chk.process_batch(xs, np.full(lead + K, pitch_value, np.float32))[:, lead * hop:]
    chk.close()
    # (with lo - lead > 0 the lead-in itself starts from an imported input tail with acc = 0: its first R - 1 hops are a halo, discarded here)
    span_equal = bool(np.array_equal(got, ref)) if lead >= R - 1 or lo == 0 else None
    import_span_state()
    with torch.cuda.stream(stream):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for _ in range(steps):
            step()
        ev1.record(stream)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / steps
    elapsed, kernel_ms = shard.reduce_max([elapsed, kernel_ms], dist, dev)
    ok = shard.reduce_max([0.0 if span_equal in (True, None) else 1.0], dist, dev)[0] == 0.0
    info = pv.info()
    pv.close()
    return {"elapsed": elapsed, "kernel_ms": kernel_ms, "info": info, "span": [lo, hi], "halo_hops": lo - start, "spans_bit_exact": ok,
            "alg_bytes": nch * (hi - lo) * 2 * hop * 4, "achieved_gbs": nch * (hi - lo) * 2 * hop * 4 / (kernel_ms * 1e-3) / 1e9}


def latency_histogram(phaze_amd, fft, hop, nch, calls, local_rank, sweep, flags=0, fs=96000.0):
    """Streaming form (one render quantum per call): per-call wall latency of pv_process through the C ABI."""
    import ctypes as C
    import numpy as np
    import signals as S
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1, device_id=local_rank, flags=flags)
    L = pv._L
    x = np.stack([S.make_signal("tonal", c, 64 * hop) for c in range(nch)])
    fpt = C.POINTER(C.c_float)
    outs = [np.zeros(hop, np.float32) for _ in range(nch)]
    op = (fpt * nch)(*[o.ctypes.data_as(fpt) for o in outs])
    blocks = [[np.ascontiguousarray(x[c, m * hop:(m + 1) * hop]) for c in range(nch)] for m in range(64)]
    ips = [(fpt * nch)(*[b.ctypes.data_as(fpt) for b in blocks[m]]) for m in range(64)]
    lat = np.empty(calls, np.float64)
    for m in range(calls + 30):
        pf = (0.5 + 1.5 * ((m % 64) / 63.0)) if sweep else 1.5        # the config's sweep, one value per hop
        t0 = time.perf_counter_ns()
        rc = L.pv_process(pv._h, ips[m % 64], op, nch, hop, C.c_float(pf))
        t1 = time.perf_counter_ns()
        assert rc == 0
        if m >= 30:
            lat[m - 30] = (t1 - t0) * 1e-3
    pv.close()
    return {"p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max()), "calls": calls,
            "form": f"pv_process through the C ABI (ctypes), {nch}-ch {fft}/{hop} @ {fs / 1000:g} kHz, pitchFactor " + ("swept per hop" if sweep else "1.5") + ", one hop per call "
                    + ("(resident kernel, PV_FLAG_PERSISTENT_STREAM: quantum handed over through the BAR, no launch)" if flags & 32 else
                       "(one launch per quantum, completion words in pinned memory)"), "realtime_budget_us": hop / fs * 1e6}


def r4_(v):
    return None if v is None else float(f"{v:.4g}")


def spawn_ranks(args, n):
    """`python bench.py --gpus N` without a launcher: start the N ranks through torch.distributed.run (one process per GPU)."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:]]
    env = dict(os.environ)
    env["PHAZE_BENCH_REQUESTED_GPUS"] = str(args.gpus)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--fft", type=int, default=1024)
    ap.add_argument("--hop", type=int, default=256)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--hops", type=int, default=1 << 20, help="hops (frames per channel) per step")
    ap.add_argument("--pitch", type=float, default=1.5)
    ap.add_argument("--pitch-sweep", action="store_true", help="pitchFactor swept 0.5->2.0 per hop, period 64 hops (BASELINE configs[4]'s schedule) instead of --pitch")
    ap.add_argument("--frames-per-chunk", type=int, default=0)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps launches each; the line reports the median region and lists all (min / median / max)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend under torch.distributed.run.  nccl (= RCCL) is what the driver runs; gloo exists for ONE test that puts two ranks on one "
                         "GPU (RCCL refuses that) to execute the multi-rank code paths end to end (tests/test_gpu_multirank.py)")
    ap.add_argument("--preheat-seconds", type=float, default=-1.0,
                    help="untimed launches of the headline workload in front of its --warmup steps, until the GPU has been busy this long.  Default: 0 when the other "
                         "configurations run first (N = 1: they already keep the box busy for ~20 s) and with --no-extras (profiling runs); 8 for N > 1, where the headline is "
                         "all a rank runs: a freshly leased box spends its first seconds below its sustained clocks, and the per-N values must be measured in the same state")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-config lines and the latency histogram (profiling runs)")
    ap.add_argument("--allow-lib-override", action="store_true", help="accept PHAZE_LIB (A/B builds of the same ABI); recorded in the line")
    ap.add_argument("--pcie", action="store_true", help="also report the host-buffer (PCIe-inclusive) rate")
    ap.add_argument("--time-shard", action="store_true",
                    help="ONE stream of --hops hops split along the time axis over the ranks (strong scaling; single-stream configs C2 / C3 on N GPUs)")
    ap.add_argument("--simulate-shard", default="", help="with --time-shard on ONE GPU: measure the span of rank r of w ('r/w') without a process group")
    ap.add_argument("--sg-streams", type=int, default=0, help="with --scatter-gather: number of streams rank 0 scatters (default: one per rank; a count the ranks "
                                                              "do not divide takes the ragged send / recv branch)")
    ap.add_argument("--scatter-gather", action="store_true",
                    help="N > 1: all streams start on rank 0, are scattered over RCCL, and the results gathered back (reported separately, never in value)")
    args = ap.parse_args()

    # the library reads no environment; PHAZE_* variables only exist for this harness (PHAZE_LIB = A/B build, explicit opt-in)
    stray = sorted(k for k in os.environ if k.startswith("PHAZE_") and k not in ("PHAZE_LIB", "PHAZE_BENCH_REQUESTED_GPUS", "PHAZE_NO_TORCH_PRELOAD"))
    if stray:
        raise SystemExit(f"bench.py refuses to run with {stray} set: the product has no environment switches")
    if os.environ.get("PHAZE_LIB") and not args.allow_lib_override:
        raise SystemExit("PHAZE_LIB is set: pass --allow-lib-override to bench a non-default build (it is recorded in the output)")

    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    requested = int(os.environ.get("PHAZE_BENCH_REQUESTED_GPUS", args.gpus))
    if "RANK" not in os.environ and args.gpus > 1:
        # self-spawn: one rank per available GPU; fewer devices than requested -> measure what exists and say so
        n = min(args.gpus, ndev)
        if n > 1:
            sys.exit(spawn_ranks(args, n))
        os.environ["PHAZE_BENCH_REQUESTED_GPUS"] = str(args.gpus)
        requested = args.gpus

    import phaze_amd
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:          # under torch.distributed.run (also at world size 1: exercises RCCL)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    fft, hop, nch, T = args.fft, args.hop, args.channels, args.hops
    from phaze_amd import shard
    if args.time_shard:
        sr, sw = rank, world
        if args.simulate_shard and world == 1:
            sr, sw = (int(v) for v in args.simulate_shard.split("/"))
        r = measure_time_shard(torch, phaze_amd, dev, dist, fft, hop, nch, T, args.pitch, args.steps, args.warmup, local_rank, sr, sw)
        if args.simulate_shard and world == 1:
            r["elapsed"] *= 1.0        # one span measured; the line below still divides the WHOLE stream by it: only meaningful as a per-span time
        if rank == 0:
            chs = "mono" if nch == 1 else "stereo" if nch == 2 else f"{nch}-ch"
            print(json.dumps({
                "metric": "stft_frames_per_sec_1024pt_hop256_48k" if (fft, hop) == (1024, 256) else f"stft_frames_per_sec_{fft}pt_hop{hop}",
                "value": nch * T * args.steps / r["elapsed"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": r["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE,
                "data": "synthetic",
                "config": {"workload": f"ONE {chs} 48 kHz stream, FFT={fft} hop={hop} pitchFactor={args.pitch}, {T} hops per step split along the time axis "
                                       f"over {world} GPU(s): every span imports its input tail + acc = 0 + timeCursor {fft // hop - 1} hops early and "
                                       "recomputes the halo (pv_import_state; no hand-over between ranks, no collective)",
                           "fft": fft, "hop": hop, "channels": nch, "hops_per_step": T, "pitch_factor": args.pitch, "rank0_span": r["span"],
                           "halo_hops": r["halo_hops"], "parallelism": f"time-shard x{world}" + (f" (simulated span {args.simulate_shard}: value is NOT a job rate)" if args.simulate_shard else ""),
                           "device": r["info"]["device_name"]},
                "roofline": {"bound": "hbm", "achieved": r["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["achieved_gbs"] / HBM_PEAK_GBS,
                             "traffic": None, "kernel": r["info"]["kernel_name"], "kernel_ms": r["kernel_ms"], "algorithmic_bytes_per_launch": r["alg_bytes"]},
                "span_starts_bit_exact_vs_processed_lead_in": r["spans_bit_exact"]}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    pitch = torch.full((T,), args.pitch, device=dev, dtype=torch.float32)
    if args.pitch_sweep:
        pitch = (0.5 + 1.5 * (torch.arange(T, device=dev) % 64).to(torch.float32) / 63.0).to(torch.float32)
        args.pitch_num, args.pitch = 1.0, "swept 0.5->2.0 per hop (period 64 hops)"    # (CPU baseline / PCIe legs of a swept run use f = 1.0)
    else:
        args.pitch_num = args.pitch
    # The other configurations are measured FIRST and the headline LAST (round 6): a freshly leased box spends its first seconds below its sustained clocks -- on some boxes the
    # first five or six workloads of a process read 3-5 % slow (the headline 1.78 ms where the same box gives 1.69 once it has been busy for ten seconds), whatever their own
    # warm-up steps -- and `value` is the steady-state rate.  Nothing about the headline's own measurement changes: W untimed steps, then regions of exactly K timed steps.
    xo, xd = {}, {}
    if world == 1 and not args.no_extras:
        # ---- the other configurations, each a short measured entry (same harness, same timing method).  Ids: profiles/bench_workloads.md.
        #      BASELINE's product configurations first, then the unfavourable parameter range, then flavours / signal classes ----
        extras, extras_full = [], []
        r4 = lambda v: None if v is None else float(f"{v:.4g}")
        def add(cid, f2, h2, c2, T2, pt, steps=8, warm=3, **kw):
            r = measure(torch, phaze_amd, dev, None, f2, h2, c2, T2, pt, steps, warm, cid, local_rank, parity_hops=12, **kw)
            e = {"id": cid, "v": r4(r["frames_per_step_rank"] * steps / r["elapsed"]), "frac": r4(r["achieved_gbs"] / HBM_PEAK_GBS), "ms": r4(r["kernel_ms"]), "par": r4(r["parity"])}
            if r["fallback_rate"] is not None:
                e["fb"] = r4(r["fallback_rate"])
            extras.append(e)
            extras_full.append(dict(e, kernel=r["info"]["kernel_name"], steps=steps, warmup=warm, ms_per_step=r["elapsed"] / steps * 1e3, frames_per_chunk=r["info"]["frames_per_chunk"],
                                    fft=f2, hop=h2, channels=c2, hops=T2))
        full = lambda n, v: torch.full((n,), v, device=dev, dtype=torch.float32)
        swp = lambda n: (0.5 + 1.5 * (torch.arange(n, device=dev) % 64).to(torch.float32) / 63.0).to(torch.float32)
        T3, T5, T8, T2 = 1 << 18, 1 << 14, 1 << 17, 1 << 20
        add("C3", 2048, 512, 2, T3, full(T3, 0.8))
        add("C4-share", 4096, 1024, 1024, 64, full(64, 1.25), steps=40, warm=10, ch_per_stream=8)   # 1 ms launches: as many steps as the headline
        add("C5-sweep", 8192, 2048, 8, T5, swp(T5))
        add("C5-f1.5", 8192, 2048, 8, T5, full(T5, 1.5))
        add("8ch-1024", 1024, 256, 8, T8, full(T8, 1.5), steps=40, warm=10)    # (as many steps as the headline: the fixed cost of a timed region read as a "gap" to mono in rounds 3-4)
        add("native", 2048, 128, 2, T3, full(T3, 1.0))
        add("C2-f0.8", 1024, 256, 1, T2, full(T2, 0.8), steps=24, warm=6)
        add("C2-sweep", 1024, 256, 1, T2, swp(T2), steps=24, warm=6)
        add("C3-f1.5", 2048, 512, 2, T3, full(T3, 1.5))
        add("C5-f0.6", 8192, 2048, 8, T5, full(T5, 0.6))
        add("N16384", 16384, 4096, 8, 1 << 11, full(1 << 11, 1.5), steps=4, warm=1)     # complete, not tuned (generic kernel with a device-memory scratch): one timing
        # the fp32-first forward transform from every side: PV_FLAG_FP64_FORWARD (the round-4 arithmetic) and the signal classes that bound it
        p15 = full(T2, 1.5)
        add("C2-fwd64", 1024, 256, 1, T2, p15, steps=24, warm=6, flags=phaze_amd.FLAG_FP64_FORWARD)
        for sig in ("white", "tonal80", "tonal60", "q16", "silence"):
            add("C2-" + sig, 1024, 256, 1, T2, p15, steps=16, warm=4, signal=sig)
            if sig != "white":
                add("C2-" + sig + "-fwd64", 1024, 256, 1, T2, p15, steps=16, warm=4, signal=sig, flags=phaze_amd.FLAG_FP64_FORWARD)
        add("C3-tonal80", 2048, 512, 2, T3, full(T3, 0.8), signal="tonal80")
        add("C3-tonal80-fwd64", 2048, 512, 2, T3, full(T3, 0.8), signal="tonal80", flags=phaze_amd.FLAG_FP64_FORWARD)
        # ---- the reference-width flavour (never the product; build/exp/libphaze_fp64.so, `make -C phaze_amd/csrc fp64`): shifted spectrum, scatter, residue, c2r pass
        #      and inverse FFT in fp64 like the reference's JS doubles.  A library is chosen at import time, so it runs in a child ----
        flib = os.path.join(ROOT, "build", "exp", "libphaze_fp64.so")
        if os.path.exists(flib) and not os.environ.get("PHAZE_LIB"):
            for cid, fargs in (("C2-f64", ["--pitch", "1.5"]), ("C2-f0.8-f64", ["--pitch", "0.8"]),
                               ("C3-f64", ["--fft", "2048", "--hop", "512", "--channels", "2", "--hops", "262144", "--pitch", "0.8"]),
                               ("C4-share-f64", ["--fft", "4096", "--hop", "1024", "--channels", "1024", "--hops", "64", "--pitch", "1.25"]),
                               ("C5-sweep-f64", ["--fft", "8192", "--hop", "2048", "--channels", "8", "--hops", "16384", "--pitch-sweep"]),
                               ("C5-f1.5-f64", ["--fft", "8192", "--hop", "2048", "--channels", "8", "--hops", "16384", "--pitch", "1.5"])):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-extras", "--no-cpu-baseline", "--allow-lib-override", "--steps", "10", "--warmup", "3",
                                    "--repeats", "3"] + fargs, capture_output=True, text=True, timeout=600, env=dict(os.environ, PHAZE_LIB=flib))
                try:
                    fj = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                    e = {"id": cid, "v": r4(fj["value"]), "frac": r4(fj["roofline"]["frac"]), "ms": r4(fj["roofline"]["kernel_ms"]), "par": r4(fj["parity_rms_vs_oracle"])}
                    extras.append(e)
                    extras_full.append(dict(e, dtype="f64", kernel=fj["roofline"]["kernel"], lib="build/exp/libphaze_fp64.so", steps=fj["steps"], ms_per_step=fj["ms_per_step"]))
                except Exception as ex:
                    extras.append({"id": cid, "err": f"{ex}"[:60]})
                    extras_full.append({"id": cid, "error": f"{ex}: {(r.stderr or r.stdout)[-300:]}"})
        xo["configs"] = extras
        xd["configs"] = extras_full
        # ---- the product boundary with HOST pointers: what a Node / C caller that owns host memory gets, PCIe included.  Never `value`; each entry carries the
        #      fraction of this box's best pinned hipMemcpy rate it reaches ----
        bw = pcie_bandwidth(torch, dev)
        hb = []
        rows4 = np.stack([np.full(64, 1.25, np.float32) for _ in range(128)])
        hb.append(host_batch(torch, phaze_amd, dev, 4096, 1024, 1024, 64, 8, rows4, 5, local_rank, bw, "C4-share-host"))
        hb.append(host_batch(torch, phaze_amd, dev, 1024, 256, 8, 1 << 15, 8, np.full((1, 1 << 15), 1.5, np.float32), 5, local_rank, bw, "8ch-1024-host"))
        nl = node_sharded_line(bw, 128, 8, 4096, 1024, 64, 5)
        if nl:
            nl["workload"] = "C4-share-node"
            hb.append(nl)
        xd["host_buffer_configs"] = hb
        xo["host"] = [{"id": h["workload"], "v": r4(h.get("value")), "gbs_each_way": r4(h.get("gbytes_per_s_each_way")), "pcie_frac": r4(h.get("pcie_frac")),
                        **({"bit_equal": h["bit_equal_to_resident_form"]} if "bit_equal_to_resident_form" in h else {}), **({"err": h["error"][:60]} if "error" in h else {})} for h in hb]
        xo["pcie_pinned_gbs"] = r4(bw["reference"])
        lat = {"C5-launch": latency_histogram(phaze_amd, 8192, 2048, 8, 300, local_rank, sweep=True),
               "C5-resident": latency_histogram(phaze_amd, 8192, 2048, 8, 300, local_rank, sweep=True, flags=32),   # PV_FLAG_PERSISTENT_STREAM
               "C2-launch": latency_histogram(phaze_amd, 1024, 256, 1, 1000, local_rank, sweep=False, fs=48000.0),
               "C2-resident": latency_histogram(phaze_amd, 1024, 256, 1, 1000, local_rank, sweep=False, flags=32, fs=48000.0)}
        xd["latency_us"] = lat
        xo["latency_us"] = {k: {"p50": r4(v["p50"]), "p99": r4(v["p99"]), "budget": r4(v["realtime_budget_us"])} for k, v in lat.items()}

    preheat = args.preheat_seconds if args.preheat_seconds >= 0 else (8.0 if world > 1 else 0.0)
    head = measure(torch, phaze_amd, dev, dist, fft, hop, nch, T, pitch, args.steps, args.warmup, "headline", local_rank,
                   frames_per_chunk=args.frames_per_chunk, parity_hops=96 if rank == 0 else 0, seed=rank, repeats=args.repeats,
                   keep_prefix_hops=(1 << 19) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else 0, preheat_s=preheat)

    sg_ms = None
    if args.scatter_gather and dist is not None:             # also at world size 1 under torchrun: the collectives still run through RCCL
        # the only exchange a multi-GPU job can have: whole streams out from rank 0 and results back (outside the timed region)
        nsg = args.sg_streams if args.sg_streams > 0 else world
        x_all = (torch.stack([synth_input(torch, nch, T * hop, dev, seed=r) for r in range(nsg)]) if rank == 0
                 else torch.empty((0, nch, T * hop), device=dev, dtype=torch.float32))
        torch.cuda.synchronize(); dist.barrier(); t_sg = time.perf_counter()
        xs = shard.scatter_streams(x_all, nsg, dist)
        ys = shard.gather_streams(xs, nsg, dist)
        torch.cuda.synchronize(); dist.barrier()
        sg_ms = shard.reduce_max([(time.perf_counter() - t_sg) * 1e3], dist, dev)[0]
        sg_ok = None
        if rank == 0:                                                     # the round trip returns every stream to its place, untouched
            sg_ok = bool(ys.shape == x_all.shape and torch.equal(ys, x_all))
        sg_mine = len(shard.stream_partition(nsg, world, rank))
        assert xs.shape[0] == sg_mine
        del x_all, xs, ys

    info = head["info"]
    value = shard.aggregate_rate(head["frames_per_step_rank"] * args.steps, world, head["elapsed"])
    sha = csrc_sha16()

    def stamped(fname, key):
        """An entry of a profiles/*.json side file, only while it was measured on THIS build of the kernels (csrc_sha16 stamp)."""
        try:
            ent = json.load(open(os.path.join(ROOT, "profiles", fname))).get(key)
            return ent if ent and ent.get("csrc_sha16") == sha else None
        except Exception:
            return None
    shape_key = f"{fft}/{hop}/ch{nch}/hops{T}"
    tent = stamped("hbm_traffic.json", shape_key)
    traffic = tent["bytes_per_launch"] if tent else None

    out = detail = None
    if rank == 0:
        is_c2 = (fft, hop, nch) == (1024, 256, 1) and not args.pitch_sweep and abs(args.pitch - 1.5) < 1e-9
        wl = "C2" if is_c2 else f"{nch}ch-{fft}/{hop}-f{args.pitch if not args.pitch_sweep else 'sweep'}"
        kms = head["regions_kernel_ms"]
        # the device-copy ceiling of THIS box, measured in this run ("quote the measured copy BW beside" the 8 TB/s peak): a 1 GiB device-to-device
        # copy kernel (torch copy_), read + write bytes over HIP-event time, best of five
        copy_gbs = None
        if world == 1:
            a = torch.empty(1 << 28, dtype=torch.float32, device=dev); b2 = torch.empty_like(a)
            b2.copy_(a); torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); b2.copy_(a); e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            copy_gbs = 2 * a.numel() * 4 / (best * 1e-3) / 1e9
            del a, b2
        out = {
            "metric": "stft_frames_per_sec_1024pt_hop256_48k" if (fft, hop) == (1024, 256) else f"stft_frames_per_sec_{fft}pt_hop{hop}",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": wl + (": BASELINE configs[1], mono 48 kHz 1024/256 pf 1.5, 1 ch x 2^20 hops resident" if is_c2 and T == 1 << 20 else ""),
                       "fft": fft, "hop": hop, "channels": nch, "hops_per_step": T, "pitch_factor": args.pitch,
                       "frames_per_chunk": info["frames_per_chunk"], "parallelism": f"streams x{world}, no collective", "device": info["device_name"][:40]},
            "roofline": {"bound": "hbm", "achieved": head["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": head["achieved_gbs"] / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": info["kernel_name"], "kernel_ms": head["kernel_ms"], "kernel_ms_min": min(kms), "kernel_ms_max": max(kms), "regions": len(kms),
                         "algorithmic_bytes_per_launch": head["alg_bytes"], "copy_gbs_measured": copy_gbs,
                         "frac_of_measured_copy": (head["achieved_gbs"] / copy_gbs) if copy_gbs else None,
                         "note": "bound by VALU issue, not HBM: see roofline_valu"},
            "parity_rms_vs_oracle": head["parity"], "fallback_rate": head["fallback_rate"],
            "busy_before_headline": ("other configs (~20 s)" if xo else f"preheat {preheat:g} s" if preheat > 0 else "none"),
        }
        # what actually binds (round-5 verdict, item 5c): VALU issue cycles of the dominant kernel over the SIMD cycles of its launch, from the PMC pass of
        # profiles/run_profile_r06.sh (profiles/valu_issue.json, stamped with the hash of the kernel sources like hbm_traffic.json)
        vent = stamped("valu_issue.json", shape_key)
        out["roofline_valu"] = ({"bound": "valu_issue", "frac": vent["frac"], "valu_insts_per_frame": vent["valu_insts_per_frame"], "wait_any_frac": vent.get("wait_any_frac"),
                                 "src": "profiles/valu_issue.json"} if vent else None)
        cent = stamped("chain_tail.json", shape_key)          # slowest chain of the launch over the mean chain (stamps build, tools/chain_times.py)
        out["tail_over_mean"] = cent["max_over_mean"] if cent else None
        distinct = min(world, ndev)                                      # LOCAL_RANK % ndev: more ranks than visible devices share them
        if requested != world or distinct != world:
            out["n_gpus"] = distinct
            out["requested_gpus"] = max(requested, world)
            out["replicas_measured"] = distinct
            out["ranks"] = world
            out["note_gpus"] = (f"{max(requested, world)} GPUs requested, {ndev} visible: {distinct} replica(s) measured"
                                + (f" ({world} ranks share them: `value` is what those devices delivered together, NOT a {world}-GPU rate)" if distinct != world else "")
                                + "; independent shards, nothing extrapolated; no scaling curve was measured")
        if os.environ.get("PHAZE_LIB"):
            out["lib_override"] = os.environ["PHAZE_LIB"]
        if sg_ms is not None:
            out["scatter_gather_ms"] = sg_ms        # one step's input out + back over RCCL, outside the timed region
            out["scatter_gather"] = {"streams": nsg, "branch": "scatter / gather (equal shares)" if nsg % world == 0 else "send / recv (ragged shares)",
                                     "round_trip_intact": sg_ok, "rank0_streams": sg_mine}
        if dist is not None:
            out["dist_backend"] = dist.get_backend()          # "nccl" = RCCL on ROCm: barrier, all_reduce(MAX) of the timing, scatter / gather
        out.update(xo)
        detail = dict(out)
        detail.update(xd)
        detail["measured_last"] = bool(xo)
        detail["timed_regions"] = {"count": len(kms), "steps_each": args.steps, "reported": "median region", "ms_per_step": head["regions_ms_per_step"], "kernel_ms": kms}
        detail["csrc_sha16"] = sha

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline(fft, hop, args.pitch_num, head["x_prefix"], "the bench input")
            detail["cpu_baseline"] = cb
            out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": 1, "kind": "port", "sample": cb["sample"][:110], "host_cpus": cb["host_cpus"],
                                   "cpu_model": cb["cpu_model"][:40], "reference_ratio": r4_(cb.get("reference_ratio")), "reference_frames_per_s_est": r4_(cb.get("reference_frames_per_s_est"))}
        if args.pcie:
            import signals as S
            Tp = min(T, 1 << 14)
            xh = np.stack([S.make_signal("tonal", c, Tp * hop) for c in range(nch)])
            pv2 = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=Tp, device_id=local_rank)
            pv2.process_batch(xh, np.full(Tp, args.pitch_num, np.float32))
            t1 = time.perf_counter()
            for _ in range(3):
                pv2.process_batch(xh, np.full(Tp, args.pitch_num, np.float32))
            out["pcie_inclusive_frames_per_s"] = 3 * nch * Tp / (time.perf_counter() - t1)
            pv2.close()
        out["legend"] = "profiles/bench_workloads.md"
        line = json.dumps(out, separators=(",", ":"))
        while len(line) > LINE_BUDGET and out.get("configs"):            # (never expected: the entries are sized for ~5 KB) drop flavour entries from the end, say so
            out["configs"].pop(); out["truncated"] = True
            line = json.dumps(out, separators=(",", ":"))
        try:                                                              # the unabridged record (workload prose, every region, forms of the latency lines) next to the line
            ddir = os.path.join(ROOT, "gpurun_out")
            os.makedirs(ddir, exist_ok=True)
            json.dump(detail, open(os.path.join(ddir, "bench_detail.json"), "w"), indent=1)
        except Exception:
            pass
        print(line, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
