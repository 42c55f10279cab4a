"""The restructured algorithm the HIP kernels implement (tests/gpu_algo_model.py) vs the reference goldens.

CPU-only.  Proves the packed-FFT / residue / owner-rule / root-of-unity restructuring is equivalent to the
reference semantics before it is written as kernels.  Tolerance is the north-star parity bar (1e-4 RMS);
observed errors are ~1e-7 (fp32 inverse).
"""
import numpy as np
import pytest

import gpu_algo_model as G
import signals as S

MAN = S.load_manifest()
CASES = {c["name"]: c for c in MAN["cases"] if not c.get("events") and c["fft"] <= 2048}


@pytest.mark.parametrize("name", sorted(CASES))
def test_model_matches_reference(name):
    case = CASES[name]
    N, h = case["fft"], case["hop"]
    T = min(case["store_hops"], 24)
    pitch = S.pitch_schedule(case["pitch"], case["nhops"])
    gold = S.load_golden_out(case)
    for ch in range(min(case["store_ch"], 1)):
        x = S.make_signal(case["signal"], ch, case["nhops"] * h)
        m = G.Model(N, h)
        y = np.concatenate([m.process(x[i * h:(i + 1) * h], pitch[i]) for i in range(T)])
        err = S.rms(y.astype(np.float64) - gold[ch, :T * h])
        assert err < 2e-7, f"{name} ch{ch}: rms err {err:.3e}"


@pytest.mark.parametrize("name", sorted(n for n, c in CASES.items() if c.get("dumps")))
def test_model_intermediates(name):
    case = CASES[name]
    N, h = case["fft"], case["hop"]
    pitch = S.pitch_schedule(case["pitch"], case["nhops"])
    x = S.make_signal(case["signal"], 0, case["nhops"] * h)
    m = G.Model(N, h)
    want = {d["hop"]: S.load_dump(case, d) for d in case["dumps"]}
    for i in range(max(want) + 1):
        m.process(x[i * h:(i + 1) * h], pitch[i])
        if i in want:
            ref, got = want[i], m.last
            Xr = ref["X"][0::2] + 1j * ref["X"][1::2]
            scale = np.max(np.abs(Xr))
            assert np.max(np.abs(got["X"] - Xr[:N // 2 + 1])) < 1e-12 * scale
            assert np.array_equal(got["peaks"], ref["peaks"])
            if got["res"] is not None:
                assert np.max(np.abs(got["res"][N // 2 + 1:] - Xr[N // 2 + 1:])) < 2e-6 * scale
            Yr = ref["Y"][0::2] + 1j * ref["Y"][1::2]
            # bins 0 and N/2: imaginary parts are dropped by the real-part extraction, compare real only there
            assert np.max(np.abs(got["Y"][1:-1] - Yr[1:-1])) < 2e-6 * scale
