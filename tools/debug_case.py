"""Debug aid: per-hop error of a golden case + intermediate comparison against the oracle (GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, signals as S, oracle_lib, phaze_amd
name = sys.argv[1]
case = {c["name"]: c for c in S.load_manifest()["cases"]}[name]
N, h, T = case["fft"], case["hop"], case["store_hops"]
x = S.make_signal(case["signal"], 0, case["nhops"] * h)
p = S.pitch_schedule(case["pitch"], case["nhops"])
pv = phaze_amd.PhaseVocoder(fft_size=N, hop_size=h, max_channels=1, max_hops=T)
o = oracle_lib.Oracle(N, h, 1)
H = N // 2 + 1
for m in range(T):
    blk = x[m * h:(m + 1) * h]
    d = pv.debug_frame(0, blk, p[m])
    yo = o.process([blk], p[m])[0]
    od = o.debug()
    outs = [[np.zeros(h, np.float32)]]
    pv.process([[blk]], outs, {"pitchFactor": np.array([p[m]], np.float32)})
    err = S.rms(outs[0][0].astype(np.float64) - yo)
    Yo = (od["Y"][0:2 * H:2] + 1j * od["Y"][1:2 * H:2])
    Yg = d["Y"][0::2] + 1j * d["Y"][1::2]
    Xo = od["X"][0::2] + 1j * od["X"][1::2]
    Xg = d["X"][0::2] + 1j * d["X"][1::2]
    pk = np.nonzero(d["flags"])[0]
    bad = np.nonzero(np.abs(Yg[1:-1] - Yo[1:-1]) > 1e-4 * np.max(np.abs(Xo)))[0] + 1
    print(f"hop {m} pf={p[m]:.3f} out_err={err:.2e} peaks_equal={np.array_equal(pk, od['peaks'])} Xerr={np.max(np.abs(Xg[:H]-Xo[:H])):.1e} "
          f"Yerr={np.max(np.abs(Yg[1:-1]-Yo[1:-1])):.2e} nbadY={len(bad)} first_bad={bad[:6]} res_err={np.max(np.abs(Xg[H:]-Xo[H:])) if np.any(Xg[H:]!=0) else -1:.2e}")
    if len(bad) and m in (1,):
        for k in list(bad[:4]) + list(bad[-3:]):
            print("   k", k, "gpu", Yg[k], "ref", Yo[k], "ratio", Yg[k] / Yo[k] if Yo[k] != 0 else None)
        lp = int(od["peaks"][-1]); x_ = lp * float(np.float32(p[m])); psh = np.floor(x_ + 0.5); print("   last peak", lp, "psh", psh, "delta", psh - lp, "upper_end", 513 - (psh - lp))
