"""Pins the CPU oracle (oracle/pv_oracle.c) against every golden fixture generated from the reference JS.

CPU-only (`-m "not gpu"`).  Tolerance: the oracle follows the reference arithmetic operation for
operation; the only difference is libm vs V8 cos/sin (<= 1 ulp fp64), so outputs agree to ~1e-8 RMS.
"""
import numpy as np
import pytest

import oracle_lib
import signals as S

MAN = S.load_manifest()
CASES = {c["name"]: c for c in MAN["cases"]}
ORACLE_TOL_RMS = 2e-8      # absolute RMS of (oracle - reference); signals are O(0.1) RMS


def _inputs(case):
    nmax = S.case_max_channels(case)
    sig = [S.make_signal(case["signal"], ch, case["nhops"] * case["hop"]) for ch in range(nmax)]
    return sig, S.pitch_schedule(case["pitch"], case["nhops"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_signal_generators_match_js(name):
    case = CASES[name]
    sig, pitch = _inputs(case)
    assert S.sha256_hex(sig[0]) == case["in_sha256_ch0"]
    assert S.sha256_hex(pitch) == case["pitch_sha256"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_reference_output(name):
    case = CASES[name]
    sig, pitch = _inputs(case)
    out, _ = oracle_lib.run_case(case, sig, pitch)
    gold = S.load_golden_out(case)
    n = case["store_hops"] * case["hop"]
    got = out[:case["store_ch"], :n]
    assert np.all(np.isfinite(got))
    err = S.rms(got.astype(np.float64) - gold.astype(np.float64))
    assert err <= ORACLE_TOL_RMS, f"{name}: rms err {err:.3e}"
    # the stored prefix hash guards the fixture files themselves
    assert S.sha256_hex(gold) == case["out_sha256"]


@pytest.mark.parametrize("case", MAN["multi_cases"], ids=lambda c: c["name"])
def test_oracle_matches_reference_multi_input(case):
    """numberOfInputs = 2 with different channel counts, one input changing its count mid-stream (ola-processor.js:24-33,38-52)."""
    sig = S.multi_case_signals(case)
    pitch = S.pitch_schedule(case["pitch"], case["nhops"])
    assert S.sha256_hex(sig[0][0]) == case["in_sha256_ch0"] and S.sha256_hex(pitch) == case["pitch_sha256"]
    outs = oracle_lib.run_multi_case(case, sig, pitch)
    gold = S.load_golden_multi(case)
    n = case["store_hops"] * case["hop"]
    for i, g in enumerate(gold):
        err = S.rms(outs[i][:, :n].astype(np.float64) - g.astype(np.float64))
        assert err <= ORACLE_TOL_RMS, f"{case['name']} input {i}: rms err {err:.3e}"


@pytest.mark.parametrize("name", sorted(n for n, c in CASES.items() if c.get("dumps")))
def test_oracle_intermediates_match_reference(name):
    case = CASES[name]
    sig, pitch = _inputs(case)
    _, dumps = oracle_lib.run_case(case, sig, pitch, collect_dumps=True)
    for d in case["dumps"]:
        ref = S.load_dump(case, d)
        got = dumps[d["hop"]]
        N = case["fft"]
        scale = np.max(np.abs(ref["X"])) + 1e-300
        # forward spectrum INCLUDING the above-Nyquist residue: all 2N doubles
        assert np.max(np.abs(got["X"] - ref["X"])) <= 1e-12 * scale
        assert np.array_equal(got["peaks"], ref["peaks"])
        np.testing.assert_allclose(got["mag"], ref["mag"], rtol=1e-6, atol=0)
        H2 = 2 * (N // 2 + 1)
        assert np.max(np.abs(got["Y"][:H2] - ref["Y"])) <= 1e-11 * scale


def test_known_answer_identity_pf1():
    """K1: pitchFactor=1 => y[n] = 0.375*x[n-(N-hop)] (periodic Hann^2 at 4 overlaps, /R)."""
    N, h, T = 1024, 256, 40
    x = S.make_signal("noise", 0, T * h)
    o = oracle_lib.Oracle(N, h, 1)
    y = o.process_planar(x[None, :], np.ones(T, np.float32))[0]
    d = N - h
    assert S.rms(y[d:] - 0.375 * x[:-d]) < 5e-8
    assert np.all(y[:d] == 0) or S.rms(y[:d]) < 1e-1   # ramp-in region is partial windows, only bounded


def test_known_answer_silence_and_errors():
    o = oracle_lib.Oracle(1024, 256, 1)
    y = o.process_planar(np.zeros((1, 256 * 8), np.float32), np.full(8, 1.5, np.float32))
    assert np.all(y == 0)
    for bad in (0, 1, 3, 1000):
        with pytest.raises(ValueError):
            oracle_lib.Oracle(bad, 1, 1)


def test_known_answer_stereo_equals_two_monos():
    """K5: channels are bit-exactly independent."""
    N, h, T = 2048, 512, 10
    xs = np.stack([S.make_signal("tonal", c, T * h) for c in range(2)])
    p = np.full(T, 0.8, np.float32)
    ys = oracle_lib.Oracle(N, h, 2).process_planar(xs, p)
    for c in range(2):
        ym = oracle_lib.Oracle(N, h, 1).process_planar(xs[c:c + 1], p)
        assert np.array_equal(ys[c], ym[0])
