"""Stations of ONE streaming quantum inside pv_wg_kernel / pv_wave2k_kernel (measurement builds -DPV_WG_STAMPS / -DPV_W2K_STAMPS, `make variant`):
usage PHAZE_LIB=build/exp/libphaze_wgst.so python tools/read_wg_stamps.py [fft hop nch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, phaze_amd, signals as S
fft, hop, nch = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 2048, 8)
pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1)
L = pv._L
x = np.stack([S.make_signal("tonal", c, 64 * hop) for c in range(nch)])
acc = []
for m in range(200):
    blk = [np.ascontiguousarray(x[c, (m % 64) * hop:((m % 64) + 1) * hop]) for c in range(nch)]
    outs = [np.zeros(hop, np.float32) for _ in range(nch)]
    pv.process([blk], [outs], {"pitchFactor": np.array([1.5], np.float32)})
    buf = np.zeros((nch, 16), np.uint32)
    L.pv_exp_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert L.pv_exp_read_stamps(pv._h, buf.ctypes.data_as(C.c_void_p), nch) == 0
    if m >= 20: acc.append(np.diff(buf[:, :7].astype(np.int64), axis=1) & 0xFFFFFFFF)
pv.close()
a = np.array(acc).mean(axis=(0, 1))
names = ["tables built", "per-quantum setup + loads issued + barrier", "input rows arrived", "frame", "state written", "signal (fence + flag)"]
print("s_memtime ticks (shader clock: ~2.0 GHz while one frame is all the GPU has to do) per station, mean over channels and quanta")
for n, v in zip(names, a): print(f"  {v:8.0f}  {n}")
print("  total", a.sum())
