"""The reference-width flavour of the N = 1024 kernel and (round 5) of the N = 4096 / 8192 kernel (build/exp/libphaze_fp64.so, `make -C phaze_amd/csrc fp64`; never the
product): shifted spectrum, scatter, above-Nyquist residue, c2r pass and inverse FFT in fp64 like the reference (bundle:102-114, phase-vocoder.js:37-39,161-170), for EVERY
pitchFactor.  Run in a subprocess (the library is chosen at import time through PHAZE_LIB) against the N = 1024 goldens generated from the
reference itself and against the oracle on random cases; the bar is the flavour's own: 1e-9 RMS (the product's bar is 2e-6, its measured error ~6e-9)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "build", "exp", "libphaze_fp64.so")

WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import phaze_amd, oracle_lib, signals as S
assert phaze_amd.library_path().endswith("libphaze_fp64.so")
FFTS = [int(v) for v in sys.argv[3].split(",")]
out = {"golden": {}, "fuzz_worst": 0.0, "fuzz_cases": 0, "kernel": None}
for c in S.load_manifest()["cases"]:
    if c.get("fft") not in FFTS or c.get("events") or c.get("arate") or not c.get("store_ch") or (c["fft"] != 1024 and c["hop"] * 8 < c["fft"]):
        continue
    h, T, nch = c["hop"], c["store_hops"], c["store_ch"]
    sig = np.stack([S.make_signal(c["signal"], ch, c["nhops"] * h) for ch in range(nch)])
    pitch = S.pitch_schedule(c["pitch"], c["nhops"])
    gold = S.load_golden_out(c)
    pv = phaze_amd.PhaseVocoder(fft_size=c["fft"], hop_size=h, max_channels=nch, max_hops=T)
    y = pv.process_batch(sig[:, :T * h], pitch[:T])
    out["kernel"] = pv.info()["kernel_name"]
    pv.close()
    out["golden"][c["name"]] = float(S.rms(y.astype(np.float64) - gold))
rng = np.random.default_rng(64)
for it in range(60 if FFTS == [1024] else 36):
    fft = FFTS[it % len(FFTS)]
    hop = int(rng.choice([fft // 8, fft // 4, fft // 2, fft]))
    nch, T = int(rng.integers(1, 3)), int(rng.integers(4, 40 if fft == 1024 else 20))
    mode = it % 4
    p = (rng.uniform(0.3, 3.0, T) if mode == 0 else np.full(T, rng.choice([0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.5, 2.0])) if mode == 1
         else rng.uniform(0.35, 1.0, T) if mode == 2 else rng.choice([0.0, -1.0, 0.8, 1.2, 100.0, 1e-3], size=T)).astype(np.float32)
    x = np.stack([S.make_signal(["noise", "tonal"][it % 2], c, T * hop, stream=it) for c in range(nch)])
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T, frames_per_chunk=int(rng.choice([0, 3, 7])))
    T1 = int(rng.integers(1, T))
    y = np.concatenate([pv.process_batch(x[:, :T1 * hop], p[:T1]), pv.process_batch(x[:, T1 * hop:], p[T1:])], axis=1)
    pv.close()
    yo = oracle_lib.Oracle(fft, hop, nch).process_planar(x, p)
    assert np.all(np.isfinite(y))
    out["fuzz_worst"] = max(out["fuzz_worst"], float(S.rms(y.astype(np.float64) - yo)))
    out["fuzz_cases"] += 1
    if len(sys.argv) > 2:
        np.savez(os.path.join(sys.argv[2], f"case{it}.npz"), x=x, p=p, y=y, hop=hop, T1=T1, fft=fft)
print(json.dumps(out))
'''


@pytest.mark.skipif(not os.path.exists(LIB), reason="build/exp/libphaze_fp64.so not built (make -C phaze_amd/csrc fp64)")
def test_reference_width_flavour_matches_the_reference(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    r = subprocess.run([sys.executable, str(script), ROOT, str(tmp_path), "1024"], capture_output=True, text=True, timeout=900, env=dict(os.environ, PHAZE_LIB=LIB))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    print(j)
    assert j["kernel"] == "pv_wave_kernel_1024" and len(j["golden"]) >= 8 and j["fuzz_cases"] == 60
    worst = max(j["golden"].values())
    assert worst < 1e-9, j["golden"]
    assert j["fuzz_worst"] < 1e-9, j["fuzz_worst"]

    # ---- differential (round 5; verdict r04 item 5): the PRODUCT against the flavour on the same 60 cases.  This comparison is what found round 4's fused window
    #      multiply (3e-9: invisible to any bound against the oracle).  Measured: ~5e-9 with every forward transform in fp64 (the fp32 inverse side alone),
    #      ~1e-8 by default (guarded frames take their source spectrum from the fp32 forward transform) ----
    import numpy as np
    import phaze_amd
    import signals as S
    worst = {0: 0.0, phaze_amd.FLAG_FP64_FORWARD: 0.0}
    for it in range(60):
        d = np.load(tmp_path / f"case{it}.npz")
        x, p, yf, hop, T1 = d["x"], d["p"], d["y"], int(d["hop"]), int(d["T1"])
        for flags in worst:
            pv = phaze_amd.PhaseVocoder(fft_size=1024, hop_size=hop, max_channels=x.shape[0], max_hops=len(p), flags=flags)
            y = np.concatenate([pv.process_batch(x[:, :T1 * hop], p[:T1]), pv.process_batch(x[:, T1 * hop:], p[T1:])], axis=1)
            pv.close()
            worst[flags] = max(worst[flags], float(S.rms(y.astype(np.float64) - yf.astype(np.float64))))
    print("product vs reference-width flavour, worst rms:", worst)
    assert worst[phaze_amd.FLAG_FP64_FORWARD] < 2e-8, worst
    assert worst[0] < 4e-8, worst


@pytest.mark.skipif(not os.path.exists(LIB), reason="build/exp/libphaze_fp64.so not built (make -C phaze_amd/csrc fp64)")
def test_reference_width_flavour_at_2048_4096_and_8192(tmp_path):
    """Round 5 (verdict r04 "missing" 3 / item 6): the same flavour of pv_wg16_kernel -- C3's, C4's and C5's sizes (the flavour's sixteen-element kernel also takes N = 2048 at
    hop >= 256; hop 128 stays with pv_wave2k_kernel), every hop the kernel takes, f on both sides of 1 (claim rounds and the re-run residue in doubles), against the goldens
    generated from the reference and against the oracle; then the PRODUCT (pv_wave2k_kernel / pv_wg16_kernel) against the flavour on the same cases."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    r = subprocess.run([sys.executable, str(script), ROOT, str(tmp_path), "2048,4096,8192"], capture_output=True, text=True, timeout=1200, env=dict(os.environ, PHAZE_LIB=LIB))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    print(j)
    assert j["kernel"] == "pv_wg16_kernel" and len(j["golden"]) >= 11 and j["fuzz_cases"] == 36
    assert max(j["golden"].values()) < 1e-9, j["golden"]
    assert j["fuzz_worst"] < 1e-9, j["fuzz_worst"]
    import numpy as np
    import phaze_amd
    import signals as S
    worst = 0.0
    for it in range(36):
        d = np.load(tmp_path / f"case{it}.npz")
        x, p, yf, hop, T1, fft = d["x"], d["p"], d["y"], int(d["hop"]), int(d["T1"]), int(d["fft"])
        pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=x.shape[0], max_hops=len(p))
        y = np.concatenate([pv.process_batch(x[:, :T1 * hop], p[:T1]), pv.process_batch(x[:, T1 * hop:], p[T1:])], axis=1)
        pv.close()
        worst = max(worst, float(S.rms(y.astype(np.float64) - yf.astype(np.float64))))
    print("product vs reference-width flavour at N = 2048 / 4096 / 8192, worst rms:", worst)
    assert worst < 4e-8, worst
