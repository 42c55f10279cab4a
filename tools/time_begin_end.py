"""Where a streaming quantum's time goes on the host side: pv_process_begin (stage + launch / publish) and pv_process_end (wait + copy out) timed separately.
usage: python tools/time_begin_end.py [fft hop nch]"""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, phaze_amd, signals as S
fft, hop, nch = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 2048, 8)
for flags, label in ((0, "launch"), (32, "resident"), (16, "launch, pinned input"), (48, "resident, pinned"), (4, "launch, workgroup kernel")):
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1, flags=flags)
    L = pv._L
    x = np.stack([S.make_signal("tonal", c, 64 * hop) for c in range(nch)])
    fpt = C.POINTER(C.c_float)
    outs = [np.zeros(hop, np.float32) for _ in range(nch)]
    op = (fpt * nch)(*[o.ctypes.data_as(fpt) for o in outs])
    blocks = [[np.ascontiguousarray(x[c, m * hop:(m + 1) * hop]) for c in range(nch)] for m in range(64)]
    ips = [(fpt * nch)(*[b.ctypes.data_as(fpt) for b in blocks[m]]) for m in range(64)]
    tb, te = [], []
    for m in range(2050):
        t0 = time.perf_counter_ns(); rc = L.pv_process_begin(pv._h, ips[m % 64], nch, hop, C.c_float(1.5)); t1 = time.perf_counter_ns()
        rc2 = L.pv_process_end(pv._h, op); t2 = time.perf_counter_ns()
        assert rc == 0 and rc2 == 0
        if m >= 50: tb.append((t1 - t0) * 1e-3); te.append((t2 - t1) * 1e-3)
    pv.close()
    print(f"{fft}/{hop} x{nch} {label:24s} begin p50 {np.percentile(tb, 50):6.1f} us   end p50 {np.percentile(te, 50):6.1f} us   sum {np.percentile(np.add(tb, te), 50):6.1f}")
