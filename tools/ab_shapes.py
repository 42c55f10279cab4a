#!/usr/bin/env python
"""A/B timing of library builds on the BASELINE shapes, one process per build (PHAZE_LIB is read at import): prints one line per shape with the
HIP-event time of a resident launch and the parity of its first hops against the oracle.  Design aid (tools/ab.sh), not part of the product.
    PHAZE_LIB=build/exp/libphaze_x.so python tools/ab_shapes.py label shape[,shape...] [steps]
shapes: head headf08 headf07 headf06 headsweep c3 c3f15 c3f07 c4 c4f08 c5 c5f08 c5f07 c5f06 c5sweep native nativef08 h128 h512"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import phaze_amd
import oracle_lib

SH = {  # fft, hop, nch, T, cps, pitch
    "head": (1024, 256, 1, 1 << 20, 1, 1.5), "headf08": (1024, 256, 1, 1 << 20, 1, 0.8), "headf07": (1024, 256, 1, 1 << 20, 1, 0.7),
    "headf06": (1024, 256, 1, 1 << 20, 1, 0.6), "headf09": (1024, 256, 1, 1 << 20, 1, 0.9), "headsweep": (1024, 256, 1, 1 << 20, 1, "sweep"),
    "h128": (1024, 128, 1, 1 << 20, 1, 1.5), "h512": (1024, 512, 1, 1 << 19, 1, 1.5),
    "c3": (2048, 512, 2, 1 << 18, 2, 0.8), "c3f15": (2048, 512, 2, 1 << 18, 2, 1.5), "c3f07": (2048, 512, 2, 1 << 18, 2, 0.7), "c3sweep": (2048, 512, 2, 1 << 18, 2, "sweep"),
    "native": (2048, 128, 2, 1 << 18, 2, 1.0), "nativef08": (2048, 128, 2, 1 << 18, 2, 0.8),
    "c4": (4096, 1024, 1024, 64, 8, 1.25), "c4f08": (4096, 1024, 1024, 64, 8, 0.8),
    "c5": (8192, 2048, 8, 1 << 14, 8, 1.5), "c5f08": (8192, 2048, 8, 1 << 14, 8, 0.8), "c5f07": (8192, 2048, 8, 1 << 14, 8, 0.7),
    "c5f06": (8192, 2048, 8, 1 << 14, 8, 0.6), "c5sweep": (8192, 2048, 8, 1 << 14, 8, "sweep"),
}


def main():
    label, shapes = sys.argv[1], sys.argv[2].split(",")
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda", 0)
    for name in shapes:
        fft, hop, nch, T, cps, pf = SH[name]
        x = bench.synth_input(torch, nch, T * hop, dev, 0)
        y = torch.empty_like(x)
        pt = (0.5 + 1.5 * (torch.arange(T, device=dev) % 64).float() / 63.0) if pf == "sweep" else torch.full((T,), float(pf), device=dev)
        pt = pt.float().contiguous()
        pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1)
        st = torch.cuda.Stream(device=dev)
        pv.set_stream(st.cuda_stream)
        run = lambda: pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, pt.data_ptr(), 0, cps)
        run(); pv.synchronize()
        K = min(T, 12)
        ref = oracle_lib.Oracle(fft, hop, 1).process_planar(x[:1, :K * hop].cpu().numpy(), pt[:K].cpu().numpy())
        err = float(np.sqrt(np.mean((y[:1, :K * hop].cpu().numpy().astype(np.float64) - ref) ** 2)))
        ms = []
        with torch.cuda.stream(st):
            for _ in range(3):
                run()
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(steps):
                    run()
                e1.record(st)
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1) / steps)
        k = pv.info()["kernel_name"]
        pv.close()
        frames = nch * T
        print(f"{label:14s} {name:10s} {k:20s} ms min {min(ms):.4f} med {sorted(ms)[1]:.4f}  {frames / sorted(ms)[1] * 1e-3 / 1e6:8.2f} Mframes/s  "
              f"{frames * 2 * hop * 4 / (sorted(ms)[1] * 1e-3) / 8e12 * 100:6.2f} %  rms {err:.2e}", flush=True)
        del x, y


if __name__ == "__main__":
    main()
