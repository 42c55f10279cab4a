#!/usr/bin/env python
"""Phase clock of pv_wg16_kernel in a THROUGHPUT launch (measurement build: make -C phaze_amd/csrc variant NAME=wg16ph FILE=pv_wg16_kernel EXTRA=-DPV_WG16_PH CAPI_EXTRA=-DPV_STAMPS=1):
s_memtime deltas per phase, accumulated by wave 0 of every workgroup over its chain.  A mark behind a barrier books the wait at that barrier to the phase it closes.
usage: PHAZE_LIB=build/exp/libphaze_wg16ph.so python tools/read_wg16_phases.py c5|c4 [pitch | sweep]      (profiles/r05_wg16_phase_clock.md)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch, phaze_amd, bench
shape = sys.argv[1] if len(sys.argv) > 1 else "c5"
arg = sys.argv[2] if len(sys.argv) > 2 else "1.5"
fft, hop, nch, T, cps = (8192, 2048, 8, 16384, 8) if shape == "c5" else (4096, 1024, 1024, 64, 8)
dev = torch.device("cuda", 0)
x = bench.synth_input(torch, nch, T * hop, dev, 0)
y = torch.empty_like(x)
p = ((0.5 + 1.5 * (torch.arange(T, device=dev) % 64).float() / 63.0) if arg == "sweep" else torch.full((T,), float(arg), device=dev)).float().contiguous()
pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1)
for _ in range(3): pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, p.data_ptr(), 0, cps)
pv.synchronize()
info = pv.info(); nchunks = (T + info["frames_per_chunk"] - 1) // info["frames_per_chunk"]
nwg = nch * nchunks
buf = np.zeros((2 * nwg, 16), np.uint32)
pv._L.pv_exp_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert pv._L.pv_exp_read_stamps(pv._h, buf.ctypes.data_as(C.c_void_p), 2 * nwg) == 0
pv.close()
buf = buf.reshape(nwg, 32)
fr = buf[:, 20].astype(np.float64); ok = fr == fr.max()
a = (buf[ok, :20].astype(np.float64) / fr[ok, None]).mean(0)
names = ["fwd: window, pack, pass A (radix 16 + 15 twiddles, fp64)", "fwd: exchange inside groups of 16 lanes", "fwd: pass B (radix 16 + 15 twiddles)", "fwd: cross-wave exchange (3 barriers)",
         "fwd: pass C", "split pass (1 barrier) + |X|^2 (+ stash) + slide / prefetch issue + shifts", "barrier: magnitudes complete", "fast residue + peak flags + own peaks (1 barrier)",
         "nearest peaks + routes + zero Y (1 barrier)", "scatter (+ residue) (1 barrier)", "c2r pass + hand-over (1 barrier) + window / twiddle loads",
         "inv: pass C", "inv: cross-wave exchange (1 barrier)", "inv: twiddles + pass B", "inv: exchange inside groups of 16 lanes", "inv: twiddles + pass A",
         "window + overlap-add + stores", "barrier: end of frame", "", ""]
print(f"pv_wg16_kernel {shape} ({fft}/{hop}, {nch} ch x {T} hops), pitch {arg}: {int(ok.sum())} workgroups x {int(fr.max())} frames; shader-clock ticks per frame (wave 0): {a.sum():.0f}")
for nme, v in zip(names, a):
    if nme: print(f"  {v:8.0f}  {100 * v / a.sum():5.1f} %  {nme}")
r = (buf[ok, 22:30].astype(np.float64) / fr[ok, None]).mean(0)
if r.sum() > 0:
    print("  inside the general residue (ticks per frame, averaged over ALL frames): base blocks + barrier", f"{r[0]:.0f}", "| stages", " ".join(f"{v:.0f}" for v in r[1:7]), "| gather + scatter + barrier", f"{r[7]:.0f}", "| sum", f"{r.sum():.0f}")
