#!/usr/bin/env python
"""Validation of the fp32-first forward transform's guard band on the GPU (round 5; verdict r04 item 1a).

Runs the validation build (`make variant NAME=flip FILE=pv_wave_kernel EXTRA=-DPV_FLIP_COUNT CAPI_EXTRA=-DPV_FLIP_COUNT`), in which EVERY frame computes both the
fp32 and the fp64 forward transform and compares the two sets of peak flags, over signal classes x amplitudes x hops, and counts
    frames | frames the guard band sends to fp64 | frames whose two flag sets differ | of those NOT sent to fp64 (must be 0) | largest q of a differing bin
(q <= 1 is what the guard calls ambiguous: sqrt(1 / q_max) is the factor by which the band could shrink before a flip escapes).
    PHAZE_LIB=build/exp/libphaze_flip.so python tools/flip_count.py [frames-per-class] [out.json] [fft = 1024 | 2048 (FILE=pv_wave2k_kernel)]
Design / evidence aid, not part of the product."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import phaze_amd


def gen(kind, n, dev, seed, fft=1024):
    g = torch.Generator(device=dev); g.manual_seed(9000 + seed)
    i = torch.arange(n, device=dev, dtype=torch.float64)
    rnd = lambda: torch.rand(n, device=dev, generator=g)
    if kind == "bench":                       # bench.py synth_input: three partials + a -36 dB floor
        b = 2 * np.pi / 48000.0
        x = 0.25 * torch.sin(i * (b * (220.0 + 17 * (seed % 97)))) + 0.125 * torch.sin(i * (b * (1375.0 + 5 * (seed % 89)))) + 0.0625 * torch.sin(i * (b * 6857.0))
        return x.float() + (rnd() - 0.5) * (2.0 / 64)
    if kind == "white":
        return rnd() - 0.5
    if kind.startswith("tonal"):              # two sines + a noise floor at -60 / -80 / -100 dB (tools/study_fp32_decisions.py)
        db = float(kind[5:])
        x = 0.5 * torch.sin(2 * np.pi * i * (0.0123 + 1e-4 * seed)) + 0.3 * torch.sin(2 * np.pi * i * (0.0931 + 3e-4 * seed))
        return x.float() + (rnd() * 2 - 1) * 10 ** (-db / 20)
    if kind == "fuzz_noise":                  # tests/signals.py "noise": uniform, amplitude 0.5
        return (rnd() * 2 - 1) * 0.5
    if kind == "fuzz_tonal":                  # tests/signals.py "tonal": three triangle waves + a 1/64 floor
        tri = lambda P: 4.0 * torch.abs((i % P) / float(P) - 0.5) - 1.0
        return (0.25 * tri(109 + seed) + 0.125 * tri(31) + 0.0625 * tri(7)).float() + (rnd() * 2 - 1) / 64
    if kind == "sine32":                      # a partial exactly on a bin, nothing else: every other bin is rounding noise
        return (0.5 * torch.sin(2 * np.pi * (i % 32) / 32)).float()
    if kind == "impulses":                    # sparse clicks
        x = torch.zeros(n, device=dev)
        idx = torch.randint(0, n, (max(n // 3000, 1),), device=dev, generator=g)
        x[idx] = rnd()[: idx.numel()] * 2 - 1
        return x
    if kind == "chirp_am":                    # a sweep under a slow envelope with a -50 dB floor: non-stationary, levels change frame to frame
        ph = 2 * np.pi * (0.001 * i + 0.5 * (0.2 / n) * i * i)
        env = 0.5 * (1 + torch.sin(2 * np.pi * i / 30011.0))
        return (0.6 * env * torch.sin(ph)).float() + (rnd() * 2 - 1) * 10 ** (-50 / 20)
    if kind == "quantised16":                 # 16-bit material: tonal + dither rounded to 1/32768
        x = 0.4 * torch.sin(2 * np.pi * i * 0.031) + 0.2 * torch.sin(2 * np.pi * i * 0.177)
        return torch.round((x.float() + (rnd() - 0.5) / 32768) * 32768) / 32768
    # ---- adversarial classes (round 6; built to stress the error law, not drawn from audio) ----
    if kind == "adv_binpair150":              # one full-scale partial exactly on a bin + a second partial 150 dB down: every other bin is the fp32 samples' rounding noise
        return (1.0 * torch.sin(2 * np.pi * (i % 16) / 16) + 10 ** (-150 / 20) * torch.sin(2 * np.pi * i * (0.0731 + 1e-4 * seed))).float()
    if kind == "adv_nyquist":                 # Nyquist alternation +-A with a tiny tone: the largest bin is the LAST one, everything else cancels
        alt = 1.0 - 2.0 * (i % 2)
        return (0.9 * alt + 1e-4 * torch.sin(2 * np.pi * i * (0.0417 + 2e-4 * seed))).float()
    if kind == "adv_click":                   # a single huge sample per ~window in low noise: |X| flat (every comparison a near tie), max|X| = the frame's rms * sqrt(N)
        x = (rnd() * 2 - 1) * 1e-5
        idx = torch.arange(137 + 11 * seed, n, 1531, device=dev)
        x[idx] = 0.95
        return x
    if kind == "adv_halfbin":                 # partials half way between two bins, detuned by parts per million: |X|^2 of the two neighbours differ by a few ulp
        x = torch.zeros(n, device=dev, dtype=torch.float64)
        for j, (k0, a) in enumerate(((37.5, 0.4), (101.5, 0.3), (222.5, 0.2))):
            x += a * torch.sin(2 * np.pi * i * ((k0 + (seed - 3.5) * 2e-7 * (j + 1)) / float(fft)) + 0.3 * j)
        return x.float() + (rnd() * 2 - 1) * 1e-7
    raise SystemExit(kind)


ALL_KINDS = ["bench", "white", "tonal60", "tonal80", "tonal100", "fuzz_noise", "fuzz_tonal", "sine32", "impulses", "chirp_am", "quantised16",
             "adv_binpair150", "adv_nyquist", "adv_click", "adv_halfbin"]


def main():
    # flip_count.py [frames-per-class] [out.json] [fft] [log2 of the frames per run at N = 1024 (default 19)]
    per_class = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
    out_path = sys.argv[2] if len(sys.argv) > 2 else ""
    fft = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    log2t = int(sys.argv[4]) if len(sys.argv) > 4 else 19
    dev = torch.device("cuda", 0)
    L = phaze_amd.load_library()
    if not hasattr(L, "pv_exp_flip_stats"):
        raise SystemExit("this library is not the validation build: PHAZE_LIB=build/exp/libphaze_flip.so")
    L.pv_exp_flip_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    kinds = ALL_KINDS
    T = (1 << log2t) * 1024 // fft
    rows, tot, tot_cls = [], np.zeros(4, np.uint64), np.zeros(2, np.uint64)
    qmax_all = 0.0
    for kind in kinds:
        acc = np.zeros(4, np.uint64); qk = 0.0; run = 0; em = np.zeros(5); cls = np.zeros(2, np.uint64)
        while int(acc[0]) < per_class:
            hop = (fft // 4, fft // 4, fft // 8 if fft == 1024 else 128, fft // 2)[run % 4]
            amp = (1.0, 1.0, 1e-4, 1.0, 30.0, 1.0, 5e-5, 1.0)[run % 8]         # scale invariance: tiny and large signals
            pf = (1.5, 1.0, 2.0, 1.25)[run % 4]
            x = (gen(kind, T * hop, dev, run, fft) * amp)[None, :].contiguous()
            y = torch.empty_like(x)
            pt = torch.full((T,), pf, device=dev, dtype=torch.float32)
            pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=1, max_hops=1)
            pv.process_batch_device(x.data_ptr(), y.data_ptr(), 1, T, T * hop, pt.data_ptr(), 0, 1)
            st = (C.c_uint64 * 12)()
            assert L.pv_exp_flip_stats(pv._h, st) == 0
            pv.close()
            acc += np.array(st[:4], np.uint64)
            cls += np.array(st[10:12], np.uint64)
            qk = max(qk, float(np.array([st[4]], np.uint32).view(np.float32)[0]))
            em = np.maximum(em, np.array(st[5:10], np.uint32).view(np.float32).astype(np.float64))
            run += 1
            del x, y
        rows.append({"signal": kind, "frames": int(acc[0]), "guard_fallbacks": int(acc[1]), "frames_with_flag_flips": int(acc[2]), "flips_not_caught": int(acc[3]), "q_max": qk,
                     "class_b_proved_from_fp64_magnitudes": int(cls[0]), "proved_b_but_fp32_test_says_a": int(cls[1]),
                     "abs_err_beyond_8epsA_over_eps_rms": em[0], "abs_err_beyond_8epsA_over_eps_max": em[1], "abs_err_beyond_32epsA_over_eps_rms": em[2],
                     "abs_err_beyond_32epsA_over_eps_max": em[3], "max_peak_over_rms": em[4]})
        tot += acc; tot_cls += cls; qmax_all = max(qmax_all, qk)
        print(f"{kind:12s} frames {int(acc[0]):10d}  fallback {100.0 * int(acc[1]) / int(acc[0]):7.3f} %  frames with flips {int(acc[2]):9d} ({100.0 * int(acc[2]) / int(acc[0]):.3f} %)  "
              f"NOT caught {int(acc[3])}  proved B from fp64 {100.0 * int(cls[0]) / int(acc[0]):7.3f} % (inconsistent {int(cls[1])})  q_max {qk:.3e}  (err-8epsA)/(eps rms) {em[0]:.1f} /(eps max) {em[1]:.2f}  (err-32epsA)/(eps rms) {em[2]:.1f} /(eps max) {em[3]:.2f}  max/rms {em[4]:.1f}", flush=True)
    res = {"fft": fft, "frames": int(tot[0]), "guard_fallbacks": int(tot[1]), "frames_with_flag_flips": int(tot[2]), "flips_not_caught": int(tot[3]), "q_max": qmax_all,
           "class_b_proved_from_fp64_magnitudes": int(tot_cls[0]), "proved_b_but_fp32_test_says_a": int(tot_cls[1]),
           "band_shrink_margin": float(np.sqrt(1.0 / qmax_all)) if qmax_all > 0 else None, "classes": rows}
    print(json.dumps(res))
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
