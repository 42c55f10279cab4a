"""Phase clock of pv_wave2k_kernel in a THROUGHPUT launch (a measurement build with s_memtime accumulators at the priority boundaries of the frame loop -- the hooks
are not kept in the product source): usage PHAZE_LIB=build/exp/libphaze_w2kph.so python tools/read_w2k_phases.py [pitch] [hop]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch, phaze_amd
pitch = float(sys.argv[1]) if len(sys.argv) > 1 else 0.8
hop = int(sys.argv[2]) if len(sys.argv) > 2 else 512
fft, nch, T = 2048, 2, 1 << 18
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(5)
n = torch.arange(T * hop, device=dev, dtype=torch.float32)[None, :]
c = torch.arange(nch, device=dev, dtype=torch.float32)[:, None]
x = (0.25 * torch.sin(n * (0.0288 + 0.002 * c)) + 0.125 * torch.sin(n * 0.18) + (torch.rand((nch, T * hop), device=dev, generator=g) - 0.5) / 32).contiguous()
y = torch.empty_like(x); p = torch.full((T,), pitch, device=dev)
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
names = ["Hann + two forward 512-point FFTs (fp64) + DIT combine", "split pass + |X|^2 (+ stash, fast residue, shift table)", "peak flags + nearest peaks + routes", "zero Y + scatter (+ residue)",
         "c2r pre-pass + DIF split", "two inverse FFTs (packed fp32)", "Hann + overlap-add + stores"]
print(f"pv_wave2k_kernel 2048/{hop} x{nch}, pitch {pitch}: {int(ok.sum())} chains x {int(fr.max())} frames; shader-clock ticks per frame per wave: {a.sum():.0f}")
for nme, v in zip(names, a): print(f"  {v:8.0f}  {100 * v / a.sum():5.1f} %  {nme}")
