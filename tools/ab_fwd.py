#!/usr/bin/env python
"""fp32-first forward transform (default) against PV_FLAG_FP64_FORWARD (the round-4 kernels) in ONE process, per signal class: HIP-event time of a resident launch,
fallback rate (pv_forward_stats), parity of the first hops against the oracle, and the RMS difference between the two forms.  Design aid (round 5), not product.
    python tools/ab_fwd.py [shape[,shape...]] [signals] [steps]
shapes: head headf08 headsweep h128 h512 c3 c3f15 native c4 c5 ...  (tools/ab_shapes.py's table)
signals: bench,white,tonal60,tonal80,tonal100,silence"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import phaze_amd
import oracle_lib
from ab_shapes import SH


def make_signal(kind, nch, n, dev):
    import bench
    if kind == "bench":
        return bench.synth_input(torch, nch, n, dev, 0)
    g = torch.Generator(device=dev); g.manual_seed(77)
    if kind == "white":
        return (torch.rand((nch, n), device=dev, generator=g) - 0.5).float()
    if kind == "silence":
        return torch.zeros((nch, n), device=dev)
    if kind.startswith("tonal"):
        db = float(kind[5:])
        i = torch.arange(n, device=dev, dtype=torch.float64)[None, :]
        x = 0.5 * torch.sin(2 * np.pi * i * 0.0123) + 0.3 * torch.sin(2 * np.pi * i * 0.0931)
        x = x.float().expand(nch, n).clone()
        x += (torch.rand((nch, n), device=dev, generator=g) * 2 - 1) * 10 ** (-db / 20)
        return x
    raise SystemExit(f"unknown signal {kind}")


def main():
    shapes = (sys.argv[1] if len(sys.argv) > 1 else "head").split(",")
    signals = (sys.argv[2] if len(sys.argv) > 2 else "bench,white,tonal80").split(",")
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dev = torch.device("cuda", 0)
    for name in shapes:
        fft, hop, nch, T, cps, pf = SH[name]
        pt = (0.5 + 1.5 * (torch.arange(T, device=dev) % 64).float() / 63.0) if pf == "sweep" else torch.full((T,), float(pf), device=dev)
        pt = pt.float().contiguous()
        for sig in signals:
            x = make_signal(sig, nch, T * hop, dev)
            outs = {}
            for label, flags in (("fp32first", 0), ("fp64", phaze_amd.FLAG_FP64_FORWARD), ("fp32first'", 0)):
                y = torch.empty_like(x)
                pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1, flags=flags)
                st = torch.cuda.Stream(device=dev)
                pv.set_stream(st.cuda_stream)
                run = lambda: pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, pt.data_ptr(), 0, cps)
                run(); pv.synchronize()
                fr, fb = pv.forward_stats(reset=True)
                K = min(T, 16)
                ref = oracle_lib.Oracle(fft, hop, 1).process_planar(x[:1, :K * hop].cpu().numpy(), pt[:K].cpu().numpy())
                err = float(np.sqrt(np.mean((y[:1, :K * hop].cpu().numpy().astype(np.float64) - ref) ** 2)))
                ms = []
                with torch.cuda.stream(st):
                    for _ in range(2):
                        run()
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(st)
                        for _ in range(steps):
                            run()
                        e1.record(st)
                        torch.cuda.synchronize()
                        ms.append(e0.elapsed_time(e1) / steps)
                pv.close()
                med = sorted(ms)[1]
                outs[label] = y
                d = ""
                if label == "fp64":
                    a = outs["fp32first"].double(); b = y.double()
                    d = f"  rms(fp32first - fp64) {float(torch.sqrt(torch.mean((a - b) ** 2))):.2e}  max {float((a - b).abs().max()):.2e}  rms(out) {float(torch.sqrt(torch.mean(b ** 2))):.3f}"
                if label == "fp32first'":
                    d = f"  repeat bit-equal {bool(torch.equal(outs['fp32first'], y))}"
                print(f"{name:9s} {sig:9s} {label:11s} ms {min(ms):.4f}/{med:.4f}  {nch * T / med * 1e-3 / 1e6:7.2f} Mfr/s  {nch * T * 2 * hop * 4 / (med * 1e-3) / 8e12 * 100:6.2f} %  "
                      f"fallback {fb}/{fr} = {100.0 * fb / max(fr, 1):.2f} %  rms vs oracle {err:.2e}{d}", flush=True)
            del x, outs


if __name__ == "__main__":
    main()
