"""One-off long fuzz (GPU box): many random cases through every kernel variant vs the CPU oracle.  usage: python tools/fuzz_long.py <ncases> <seed> [log2n]   (log2n pins the FFT size, e.g. 10 to stress the wave kernel)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, phaze_amd, oracle_lib, signals as S
from test_gpu_fuzz import _case
n, seed = int(sys.argv[1]), int(sys.argv[2])
pin = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(seed)
worst, t0, kern = 0.0, time.time(), {}
for i in range(n):
    N, hop, nch, T, kind, p = _case(rng)
    if pin:
        N = 1 << pin; hop = max(2, N >> int(rng.integers(0, 5))); nch = int(rng.integers(1, 5)); T = len(p)
    x = np.stack([S.make_signal(kind, c, T * hop, stream=i) for c in range(nch)])
    fpc = int(rng.choice([0, 0, 1, 3, 7, 16]))
    pv = phaze_amd.PhaseVocoder(fft_size=N, hop_size=hop, max_channels=nch, max_hops=T, frames_per_chunk=fpc)
    parts, pos = [], 0
    while pos < T:
        k = int(rng.integers(1, T - pos + 1)); parts.append(pv.process_batch(x[:, pos * hop:(pos + k) * hop], p[pos:pos + k])); pos += k
    y = np.concatenate(parts, axis=1); name = pv.info()["kernel_name"]; pv.close()
    kern[name] = kern.get(name, 0) + 1
    yo = oracle_lib.Oracle(N, hop, nch).process_planar(x, p)
    err = S.rms(y.astype(np.float64) - yo)
    if not np.all(np.isfinite(y)) or err > 2e-7:
        print("FAIL", i, N, hop, nch, T, kind, fpc, name, err, p[:8]); sys.exit(1)
    worst = max(worst, err)
print("ok", n, "cases, worst rms", worst, kern, "in", round(time.time() - t0, 1), "s")
