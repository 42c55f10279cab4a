#!/usr/bin/env python
"""bench.py -- headline measurement of the phaze hot path on MI355X.

A "step" = one pass of the hot path (one pv_process_batch_device launch) over one batch of synthetic
input that is already resident in HBM.  Workload at N=1 is BASELINE.json configs[1]: mono 48 kHz,
FFT=1024, hop=256, pitchFactor=1.5, run in throughput mode: one long stream (HOPS hops per step) is
processed frame-parallel (chunks of frames with an (R-1)-frame halo).  With --gpus N every
rank owns an independent stream of the same size on its own GPU (weak scaling, no data-path collective;
RCCL is only used for the barrier / max-over-ranks reduction of the timing).

Prints ONE JSON line on rank 0 (contract in the task statement), including
  "roofline":     algorithmic HBM bytes per launch / measured launch duration vs the 8 TB/s peak
  "cpu_baseline": the CPU oracle (a C port of the reference JS, kind "port") timed on ONE host core on a
                  bounded sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


def synth_input(torch, nch, nsamples, device, seed):
    """Tonal partials + a -36 dB noise floor (always keep a noise floor, K12), generated on device."""
    g = torch.Generator(device=device)
    g.manual_seed(1234 + seed)
    n = torch.arange(nsamples, device=device, dtype=torch.float32)
    x = torch.empty((nch, nsamples), device=device, dtype=torch.float32)
    for c in range(nch):
        base = 2 * 3.14159265358979 / 48000.0
        s = 0.25 * torch.sin(n * (base * (220.0 + 17 * c))) + 0.125 * torch.sin(n * (base * (1375.0 + 5 * c))) \
            + 0.0625 * torch.sin(n * (base * 6857.0))
        s += (torch.rand(nsamples, device=device, generator=g) - 0.5) * (2.0 / 64)
        x[c] = s
    return x


def cpu_baseline(fft, hop, pitch, target_seconds=12.0):
    """Times the CPU oracle (oracle/pv_oracle.c -- the checker, used here only as the reported baseline)."""
    import numpy as np
    import oracle_lib
    import signals as S
    probe = 1024
    x = S.make_signal("tonal", 0, probe * hop)[None, :]
    p = np.full(probe, pitch, np.float32)
    o = oracle_lib.Oracle(fft, hop, 1)
    t0 = time.perf_counter()
    o.process_planar(x, p)
    dt = time.perf_counter() - t0
    n = int(max(probe, min(2000000, target_seconds / (dt / probe))))
    x = S.make_signal("tonal", 0, n * hop)[None, :]
    p = np.full(n, pitch, np.float32)
    o = oracle_lib.Oracle(fft, hop, 1)
    t0 = time.perf_counter()
    o.process_planar(x, p)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames of mono {fft}/{hop} pf={pitch} tonal+noise input, single thread, {dt:.1f} s",
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--fft", type=int, default=1024)
    ap.add_argument("--hop", type=int, default=256)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--hops", type=int, default=1 << 20, help="hops (frames per channel) per step")
    ap.add_argument("--pitch", type=float, default=1.5)
    ap.add_argument("--frames-per-chunk", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pcie", action="store_true", help="also report the host-buffer (PCIe-inclusive) rate")
    ap.add_argument("--scatter-gather", action="store_true",
                    help="N > 1: all streams start on rank 0, are scattered over RCCL, and the results gathered back (reported separately, never in value)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import phaze_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:          # under torch.distributed.run (also at world size 1: exercises RCCL)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    fft, hop, nch, T = args.fft, args.hop, args.channels, args.hops
    sg_ms = None
    if args.scatter_gather and dist is not None:
        # the only exchange a multi-GPU job can have: whole streams out from rank 0 (and results back, below)
        from phaze_amd import shard as _shard
        x_all = (torch.stack([synth_input(torch, nch, T * hop, dev, seed=r) for r in range(world)]) if rank == 0
                 else torch.empty((0, nch, T * hop), device=dev, dtype=torch.float32))
        torch.cuda.synchronize(); dist.barrier(); t_sg = time.perf_counter()
        x = _shard.scatter_streams(x_all, world, dist)[0].contiguous()
        torch.cuda.synchronize(); dist.barrier(); sg_ms = (time.perf_counter() - t_sg) * 1e3
        del x_all
    else:
        x = synth_input(torch, nch, T * hop, dev, seed=rank)
    y = torch.empty_like(x)
    pitch = torch.full((T,), args.pitch, device=dev, dtype=torch.float32)
    torch.cuda.synchronize()

    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1, device_id=local_rank,
                                frames_per_chunk=args.frames_per_chunk)
    # a real (non-null) torch stream: the library launches on it, so torch.cuda.Event brackets the kernels
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    pv.set_stream(stream.cuda_stream)

    def step():
        pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, pitch.data_ptr(), 0, 1)

    # ---- in-bench parity spot check: first hops of the resident batch vs the CPU oracle ----
    parity = None
    if rank == 0:
        import oracle_lib
        K = 96
        pv.reset()
        step()
        torch.cuda.synchronize()
        got = y[:, :K * hop].cpu().numpy()
        ref = oracle_lib.Oracle(fft, hop, nch).process_planar(x[:, :K * hop].cpu().numpy(), np.full(K, args.pitch, np.float32))
        parity = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)))
    pv.reset()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps            # HIP events on the launch stream: avg launch duration
    from phaze_amd import shard
    elapsed, kernel_ms = shard.reduce_max([elapsed, kernel_ms], dist, dev)      # MAX over ranks

    if sg_ms is not None:
        from phaze_amd import shard as _shard
        torch.cuda.synchronize(); dist.barrier(); t_sg = time.perf_counter()
        y_all = _shard.gather_streams(y.unsqueeze(0), world, dist)
        torch.cuda.synchronize(); dist.barrier(); sg_ms += (time.perf_counter() - t_sg) * 1e3
        del y_all
        sg_ms = _shard.reduce_max([sg_ms], dist, dev)[0]

    info = pv.info()
    frames_per_step_rank = nch * T
    value = shard.aggregate_rate(frames_per_step_rank * args.steps, world, elapsed)
    alg_bytes_per_launch = frames_per_step_rank * 2 * hop * 4          # 2*hop*4 B per channel-frame
    achieved = alg_bytes_per_launch / (kernel_ms * 1e-3) / 1e9          # GB/s, per GPU
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"{fft}/{hop}/ch{nch}/hops{T}"
            if key in tj:
                traffic = tj[key]["bytes_per_launch"]
        except Exception:
            traffic = None

    out = None
    if rank == 0:
        out = {
            "metric": "stft_frames_per_sec_1024pt_hop256_48k" if (fft, hop) == (1024, 256) else f"stft_frames_per_sec_{fft}pt_hop{hop}",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "dtype_note": "forward FFT and peak decisions in f64 (decision parity), shift / inverse FFT / overlap-add in f32, I/O f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: {'mono' if nch == 1 else str(nch) + '-ch'} 48 kHz FFT={fft} hop={hop} pitchFactor={args.pitch}, "
                                   f"throughput mode, one resident stream of {T} hops per GPU per step",
                       "fft": fft, "hop": hop, "channels": nch, "hops_per_step": T, "pitch_factor": args.pitch,
                       "frames_per_chunk": info["frames_per_chunk"], "threads_per_workgroup": info["threads_per_workgroup"],
                       "lds_bytes_per_workgroup": info["lds_bytes_per_workgroup"], "parallelism": f"streams x{world} (independent, no collective)",
                       "device": info["device_name"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel": info["kernel_name"], "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                         "traffic_source": "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)" if traffic else None,
                         "note": "algorithmic bytes = 2*hop*4 B per channel-frame; the kernel is LDS/VALU-bound (fp64 FFT), see DESIGN.md"},
            "parity_rms_vs_oracle": parity,
        }
        if sg_ms is not None:
            out["scatter_gather_ms"] = sg_ms        # one step's input out + output back over RCCL, outside the timed region
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(fft, hop, args.pitch)
        if args.pcie:
            xh = x[:, :min(T, 1 << 14) * hop].cpu().numpy()
            Tp = xh.shape[1] // hop
            pv2 = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=Tp, device_id=local_rank)
            pv2.process_batch(xh, np.full(Tp, args.pitch, np.float32))
            t1 = time.perf_counter()
            for _ in range(3):
                pv2.process_batch(xh, np.full(Tp, args.pitch, np.float32))
            out["pcie_inclusive_frames_per_s"] = 3 * nch * Tp / (time.perf_counter() - t1)
            pv2.close()
        print(json.dumps(out), flush=True)
    pv.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
