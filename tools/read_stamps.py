#!/usr/bin/env python
"""Phase clock of pv_wave_kernel_1024 (measurement build -DPV_STAMPS, `make variant`): one launch of the headline shape, then the per-phase
s_memtime deltas every wave accumulated over its chain.  Usage: PHAZE_LIB=build/exp/libphaze_stamps2.so python tools/read_stamps.py [pitch]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import phaze_amd

PH = ["fwd pass1 arithmetic (Hann, pack, radix-8, 7 cmul fp64)", "fwd transpose 1 (8 w128 + 8 r128)", "fwd pass2 arithmetic", "fwd transpose 2",
      "pass3 + split exchange + split arithmetic + |X|^2 + prefetch issue", "peak search -> routes", "zero Y + scatter", "c2r pre-pass + hand-over",
      "inv pass1 arithmetic (packed)", "inv transpose 1 (4 w128 + 8 r64)", "inv pass2 arithmetic", "inv transpose 2", "inv pass3 + window + overlap-add + stores"]


def main():
    pitch = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
    fft, hop, T = 1024, 256, 1 << 20
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(7)
    n = torch.arange(T * hop, device=dev, dtype=torch.float32)
    x = (0.25 * torch.sin(n * 0.0288) + 0.125 * torch.sin(n * 0.18) + (torch.rand(T * hop, device=dev, generator=g) - 0.5) / 32)[None, :].contiguous()
    y = torch.empty_like(x)
    p = torch.full((T,), pitch, device=dev, dtype=torch.float32)
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=1, max_hops=1)
    L = pv._L
    for _ in range(3):
        pv.process_batch_device(x.data_ptr(), y.data_ptr(), 1, T, T * hop, p.data_ptr(), 0, 1)
    pv.synchronize()
    info = pv.info()
    nchains = (T + info["frames_per_chunk"] - 1) // info["frames_per_chunk"]
    buf = np.zeros((nchains, 16), np.uint32)
    L.pv_exp_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    rc = L.pv_exp_read_stamps(pv._h, buf.ctypes.data_as(C.c_void_p), nchains)
    assert rc == 0, rc
    pv.close()
    frames = buf[:, 14].astype(np.float64)
    full = frames == frames.max()
    acc = buf[full, :13].astype(np.float64) / frames[full, None]
    tot = buf[full, 13].astype(np.float64) / frames[full]
    mean = acc.mean(0)
    out = {"pitch": pitch, "chains": int(full.sum()), "frames_per_chain": int(frames.max()), "ticks_per_frame_per_wave_mean": float(tot.mean()),
           "ticks_per_frame_per_wave_p5_p95": [float(np.percentile(tot, 5)), float(np.percentile(tot, 95))],
           "phases": [{"phase": PH[i], "ticks": float(mean[i]), "share": float(mean[i] / mean.sum())} for i in range(13)],
           "sum_of_phases": float(mean.sum())}
    print(json.dumps(out))
    print(f"# pitch {pitch}: {out['chains']} chains x {out['frames_per_chain']} frames; {tot.mean():.0f} ticks per frame per wave (12 waves/CU -> {tot.mean() / 12:.0f} per frame per CU)", file=sys.stderr)
    for i in range(13):
        print(f"#  {mean[i]:8.0f}  {100 * mean[i] / mean.sum():5.1f} %  {PH[i]}", file=sys.stderr)


if __name__ == "__main__":
    main()
