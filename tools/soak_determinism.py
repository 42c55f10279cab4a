#!/usr/bin/env python
"""Soak test (GPU box): repeated full-occupancy launches of one input must give bit-identical output -- a missing barrier or an LDS region reused too early shows
up as run-to-run differences long before it shows up as an error against the oracle.  usage: python tools/soak_determinism.py [repeats]   (design aid, not a pytest test)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import phaze_amd
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
dev = torch.device("cuda", 0)
bad = 0
for fft, hops_of in ((8192, (1024, 2048, 4096, 8192)), (4096, (512, 1024, 2048, 4096)), (2048, (512,)), (1024, (256,))):
    for hop in hops_of:
        for pf in (1.5, 0.8, 0.7, 0.55, "sweep", "noise"):
            nch, T = (16, 2048) if fft >= 4096 else (8, 16384)
            x = bench.synth_input(torch, nch, T * hop, dev, 3)
            if pf == "sweep":
                pt = (0.5 + 1.5 * (torch.arange(T, device=dev) % 64).float() / 63.0).float().contiguous()
            elif pf == "noise":
                g = torch.Generator(device=dev); g.manual_seed(7)
                pt = (0.4 + 2.0 * torch.rand(T, device=dev, generator=g)).float().contiguous()
            else:
                pt = torch.full((T,), float(pf), device=dev)
            pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1)
            st = torch.cuda.Stream(device=dev); pv.set_stream(st.cuda_stream)
            ref = None
            for r in range(reps):
                y = torch.empty_like(x)
                pv.reset()
                torch.cuda.synchronize()
                pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, pt.data_ptr(), 0, nch)
                pv.synchronize()
                if ref is None:
                    ref = y
                    assert torch.isfinite(y).all()
                elif not torch.equal(ref, y):
                    d = (ref != y).nonzero()
                    print("DIFF", fft, hop, pf, "run", r, "first at", d[0].tolist(), "count", d.shape[0], flush=True)
                    bad += 1
                    break
            name = pv.info()["kernel_name"]
            pv.close()
            print("ok " if bad == 0 else "-- ", fft, hop, pf, name, flush=True)
print("soak:", "clean" if bad == 0 else f"{bad} shapes differ")
sys.exit(1 if bad else 0)
