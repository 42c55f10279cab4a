import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, phaze_amd, bench
dev = torch.device("cuda", 0)
SH = {"h8": (1024, 256, 8, 1 << 17, 8, 1.5), "c3": (2048, 512, 2, 1 << 18, 2, 0.8), "c4": (4096, 1024, 1024, 64, 8, 1.25), "c5": (8192, 2048, 8, 1 << 14, 8, 1.5),
      "h64": (1024, 256, 64, 1 << 14, 8, 1.5)}
for name in sys.argv[1].split(","):
    fft, hop, nch, T, cps, pf = SH[name]
    for pad in (0, 64, 1024 + 64, 4096 + 192, 65536 + 320):
        stride = T * hop + pad
        xb = torch.zeros((nch, stride), device=dev); yb = torch.zeros((nch, stride), device=dev)
        xb[:, :T * hop] = bench.synth_input(torch, nch, T * hop, dev, 0)
        pt = torch.full((T,), float(pf), device=dev)
        pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1)
        st = torch.cuda.Stream(device=dev); pv.set_stream(st.cuda_stream)
        run = lambda: pv.process_batch_device(xb.data_ptr(), yb.data_ptr(), nch, T, stride, pt.data_ptr(), 0, cps)
        torch.cuda.synchronize(); run(); pv.synchronize()
        ms = []
        with torch.cuda.stream(st):
            for _ in range(3): run()
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(10): run()
                e1.record(st); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1) / 10)
        pv.close()
        print(f"{name} pad {pad:6d} floats: {sorted(ms)[1]:.4f} ms  {nch * T / sorted(ms)[1] / 1e3:.1f} Mframes/s", flush=True)
        del xb, yb
