"""Phase clock of pv_wg_kernel in a THROUGHPUT launch (a measurement build with s_memtime accumulators at the phase boundaries of the frame loop -- the hooks are not kept in the product source; the numbers are in profiles/r03_wg_phase_clock.md):
usage PHAZE_LIB=build/exp/libphaze_wgph.so python tools/read_wg_phases.py [pitch | sweep]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch, phaze_amd
arg = sys.argv[1] if len(sys.argv) > 1 else "1.5"
fft, hop, nch, T = 8192, 2048, 8, 16384
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
n = torch.arange(T * hop, device=dev, dtype=torch.float32)[None, :]
c = torch.arange(nch, device=dev, dtype=torch.float32)[:, None]
x = (0.25 * torch.sin(n * (0.0288 + 0.002 * c)) + 0.125 * torch.sin(n * 0.18) + (torch.rand((nch, T * hop), device=dev, generator=g) - 0.5) / 32).contiguous()
y = torch.empty_like(x)
p = ((0.5 + 1.5 * (torch.arange(T, device=dev) % 64).float() / 63.0) if arg == "sweep" else torch.full((T,), float(arg), device=dev)).float().contiguous()
pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1)
for _ in range(3): pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, p.data_ptr(), 0, 1)
pv.synchronize()
info = pv.info(); nchunks = (T + info["frames_per_chunk"] - 1) // info["frames_per_chunk"]
buf = np.zeros((nch * nchunks, 16), np.uint32)
pv._L.pv_exp_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert pv._L.pv_exp_read_stamps(pv._h, buf.ctypes.data_as(C.c_void_p), nch * nchunks) == 0
pv.close()
fr = buf[:, 8].astype(np.float64); ok = fr == fr.max()
a = (buf[ok, :8].astype(np.float64) / fr[ok, None]).mean(0)
names = ["Hann + forward FFT (fp64, 3 transposes)", "split pass + |X|^2 (+ stash)", "slide / prefetch issue + shift table + fast residue", "peak flags + nearest peaks + routes",
         "zero Y + scatter (+ residue)", "c2r pre-pass + hand-over", "inverse FFT (packed fp32)", "Hann + overlap-add + stores + barrier"]
print(f"pv_wg_kernel<13,2> C5 shape, pitch {arg}: {int(ok.sum())} workgroups x {int(fr.max())} frames; shader-clock ticks per frame: {a.sum():.0f}")
for nme, v in zip(names, a): print(f"  {v:8.0f}  {100 * v / a.sum():5.1f} %  {nme}")

