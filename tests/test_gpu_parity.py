"""GPU parity tests proper: the HIP path (through the C ABI, via ctypes) against the golden vectors generated
from the reference JS and against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): output within 1e-4 RMS (float32) of the reference.  We assert a much
tighter regression bound (observed error is fp32 round-off of the inverse FFT, ~1e-8) so that a logic
slip cannot hide inside the budget.
"""
import numpy as np
import pytest

import oracle_lib
import signals as S

pytestmark = pytest.mark.gpu

PARITY_BAR_RMS = 1e-4        # the north-star tolerance
REGRESSION_RMS = 2e-7        # what this implementation actually has to hold (round 5: was 2e-6; worst observed 8e-8 over the fuzz, typically 5e-9 .. 1.5e-8:
                             # N = 1024 / 2048 frames whose decisions the fp32 forward transform carries take their source spectrum from it, ~1e-7 of the frame's rms)

MAN = S.load_manifest()
CASES = {c["name"]: c for c in MAN["cases"]}


def _pv(**kw):
    import phaze_amd
    return phaze_amd.PhaseVocoder(**kw)


def _inputs(case):
    nmax = S.case_max_channels(case)
    sig = [S.make_signal(case["signal"], ch, case["nhops"] * case["hop"]) for ch in range(nmax)]
    return sig, S.pitch_schedule(case["pitch"], case["nhops"])


def _check(got, gold, name):
    assert np.all(np.isfinite(got)), name
    err = S.rms(got.astype(np.float64) - gold.astype(np.float64))
    assert err <= PARITY_BAR_RMS, f"{name}: rms {err:.3e} breaks the 1e-4 parity bar"
    assert err <= REGRESSION_RMS, f"{name}: rms {err:.3e} above the regression bound"
    return err


@pytest.mark.parametrize("name", sorted(n for n, c in CASES.items() if not c.get("events") and not c.get("arate")))
def test_batch_matches_reference_golden(name):
    case = CASES[name]
    sig, pitch = _inputs(case)
    T, h, nch = case["store_hops"], case["hop"], case["store_ch"]
    x = np.stack(sig[:nch])[:, :T * h]
    pv = _pv(fft_size=case["fft"], hop_size=h, max_channels=nch, max_hops=T)
    y = pv.process_batch(x, pitch[:T])
    _check(y, S.load_golden_out(case), name)
    assert pv.time_cursor == T * h
    pv.close()


@pytest.mark.parametrize("flags", [0, 16, 2 | 8, 32, 32 | 16], ids=["default", "pinned_input", "copy_nodes_event_wait", "resident", "resident_pinned_input"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_streaming_process_matches_reference_golden(name, flags):
    """process(inputs, outputs, parameters) one render quantum at a time, incl. pause / channel-change / a-rate -- in the default form (small
    quanta written by the host into device memory through the BAR, completion words), with PV_FLAG_STREAM_PINNED_INPUT (kernel reads the hop
    from pinned host memory), with PV_FLAG_STREAM_COPY | PV_FLAG_STREAM_EVENT_WAIT (copy nodes + stream wait: the round-1 form) and on the
    resident kernel (PV_FLAG_PERSISTENT_STREAM; shapes it does not cover fall back to the launch form), handed its quanta through the BAR or
    through pinned memory."""
    case = CASES[name]
    sig, pitch = _inputs(case)
    h = case["hop"]
    # every hop of the cases whose events lie late (the outputs-do-not-mirror-inputs cases: an input that disappears and returns at hops 30 .. 50) in the default form;
    # 24 hops elsewhere (the other forms run the same C-ABI entry points)
    late = any(e["hop"] >= 24 for e in case.get("events", []))
    T = case["store_hops"] if (late and flags == 0) else min(case["store_hops"], 24)
    nmax = S.case_max_channels(case)
    # cases whose outputs do not mirror their inputs ('out_channels' events: ola-processor.js:46-51 reallocates the output buffers on its own): the host does
    # the reference's bookkeeping (PV_FLAG_HOST_CHANNEL_BOOKKEEPING; phaze_amd.PhaseVocoder.process / phase-vocoder.js) with pv_reset_channels_part
    own_outputs = any(e["type"] == "out_channels" for e in case.get("events", []))
    pv = _pv(fft_size=case["fft"], hop_size=h, max_channels=nmax, max_hops=1, flags=flags | (128 if own_outputs else 0))
    out = np.zeros((nmax, T * h), np.float32)
    nch, nout = case["nch"], -1
    for m in range(T):
        paused = False
        for e in case.get("events", []):
            if e["hop"] == m:
                paused |= e["type"] == "pause"
                if e["type"] == "channels":
                    nch = e["nch"]
                if e["type"] == "out_channels":
                    nout = e["nch"]
        use = min(nch, nmax)
        inputs = [[np.zeros(0, np.float32) if paused else sig[c][m * h:(m + 1) * h] for c in range(use)]]
        outputs = [[np.zeros(h, np.float32) for _ in range(max(use, nout))]]
        pf = np.full(h, 0.7, np.float32) if case.get("arate") else np.zeros(1, np.float32)
        pf[-1] = pitch[m]
        assert pv.process(inputs, outputs, {"pitchFactor": pf}) is True
        for c in range(use):
            out[c, m * h:(m + 1) * h] = outputs[0][c]
    gold = S.load_golden_out(case)
    sc = case["store_ch"]
    _check(out[:sc], gold[:, :T * h], name)
    pv.close()


@pytest.mark.parametrize("name", sorted(n for n, c in CASES.items() if c.get("dumps")))
def test_intermediates_match_reference(name):
    """freqComplexBuffer (incl. the above-Nyquist residue when read), magnitudes, peaks, shifted spectrum -- tapped from the kernel the handle
    actually runs (pv_wave2k_kernel for the c3m_* / native_* dumps, pv_wg16_kernel for c4m_* and the 8192-point cases, each through its AUX instance)."""
    case = CASES[name]
    sig, pitch = _inputs(case)
    N, h = case["fft"], case["hop"]
    H = N // 2 + 1
    pv = _pv(fft_size=N, hop_size=h, max_channels=1, max_hops=1)
    expect = {1024: "pv_wave_kernel_1024", 2048: "pv_wave2k_kernel", 4096: "pv_wg16_kernel", 8192: "pv_wg16_kernel"}
    if N in expect and (N, h) != (1024, 64):
        assert pv.info()["kernel_name"] == expect[N]
    want = {d["hop"]: S.load_dump(case, d) for d in case["dumps"]}
    for m in range(max(want) + 1):
        blk = sig[0][m * h:(m + 1) * h]
        if m in want:
            ref, got = want[m], pv.debug_frame(0, blk, pitch[m])
            Xr = ref["X"][0::2] + 1j * ref["X"][1::2]
            Xg = got["X"][0::2] + 1j * got["X"][1::2]
            scale = np.max(np.abs(Xr))
            assert np.max(np.abs(Xg[:H] - Xr[:H])) < 1e-12 * scale, "fp64 forward spectrum"
            assert np.array_equal(np.nonzero(got["flags"])[0], ref["peaks"]), "peak set"
            np.testing.assert_allclose(got["mag"], ref["mag"], rtol=1e-6)
            Yr = ref["Y"][0::2] + 1j * ref["Y"][1::2]
            Yg = got["Y"][0::2] + 1j * got["Y"][1::2]
            assert np.max(np.abs(Yg[1:-1] - Yr[1:-1])) < 2e-6 * scale, "shifted spectrum"
            if np.any(Xg[H:] != 0):     # residue was rebuilt for this frame: compare where the reference reads it
                pk = ref["peaks"]
                if len(pk):
                    lp = int(pk[-1])
                    x = lp * float(np.float32(pitch[m]))
                    psh = np.floor(x) + (1 if x - np.floor(x) >= 0.5 else 0)
                    d = int(psh) - lp
                    if d < 0:
                        hi = min(N, H - d)
                        assert np.max(np.abs(Xg[H:hi] - Xr[H:hi])) < 2e-6 * scale, "above-Nyquist residue"
        outputs = [[np.zeros(h, np.float32)]]
        pv.process([[blk]], outputs, {"pitchFactor": np.array([pitch[m]], np.float32)})
    pv.close()


@pytest.mark.parametrize("fft,hop,pf", [(1024, 256, 1.5), (2048, 512, 0.8), (2048, 128, 1.3), (4096, 1024, 1.25), (8192, 2048, 0.9)])
def test_chunking_and_call_splitting_invariance(fft, hop, pf):
    """Frame-parallel chunks with halo == one long chain == many short calls (state carry), bit for bit: the colliding scatter (f < 1) runs
    claim rounds whose order does not depend on timing (no float atomics since round 2), so this holds for every pitch factor."""
    T = 40
    x = np.stack([S.make_signal("tonal", c, T * hop) for c in range(2)])
    p = np.full(T, pf, np.float32)
    ref = None
    for F in (T, 7, 16):
        pv = _pv(fft_size=fft, hop_size=hop, max_channels=2, max_hops=T, frames_per_chunk=F)
        y = pv.process_batch(x, p)
        pv.close()
        if ref is None:
            ref = y
        else:
            assert np.array_equal(y, ref), f"frames_per_chunk={F}"
    pv = _pv(fft_size=fft, hop_size=hop, max_channels=2, max_hops=T, frames_per_chunk=5)
    parts, pos = [], 0
    for n in (1, 3, 9, 2, 25):
        parts.append(pv.process_batch(x[:, pos * hop:(pos + n) * hop], p[pos:pos + n]))
        pos += n
    pv.close()
    assert np.array_equal(np.concatenate(parts, axis=1), ref)
    # and against the oracle
    o = oracle_lib.Oracle(fft, hop, 2)
    yo = o.process_planar(x, p)
    assert S.rms(ref.astype(np.float64) - yo) < REGRESSION_RMS


def test_oracle_parity_random_configs():
    """Seeded sweep: sizes x hops x pitch factors against the CPU oracle (sizes the oracle finishes in seconds)."""
    rng = np.random.default_rng(7)
    worst = 0.0
    for fft, hop in [(64, 16), (128, 64), (256, 32), (512, 128), (1024, 256), (1024, 64), (2048, 512), (2048, 128), (4096, 1024), (8192, 2048)]:
        for kind in ("noise", "tonal"):
            T = 12 if fft >= 4096 else 24
            nch = 2
            x = np.stack([S.make_signal(kind, c, T * hop, stream=3) for c in range(nch)])
            p = rng.uniform(0.4, 2.2, size=T).astype(np.float32)
            pv = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
            y = pv.process_batch(x, p)
            pv.close()
            yo = oracle_lib.Oracle(fft, hop, nch).process_planar(x, p)
            err = S.rms(y.astype(np.float64) - yo)
            worst = max(worst, err)
            assert err < REGRESSION_RMS, f"{fft}/{hop} {kind}: {err:.3e}"
    print("worst rms vs oracle", worst)


@pytest.mark.parametrize("fft,hop", [(512, 128), (1024, 256), (2048, 512), (4096, 512)])
def test_per_stream_pitch_rows_and_channel_independence(fft, hop):
    """Batched independent processors: channel c uses pitch row c // channels_per_stream; K5: stereo == two monos.  One size per kernel."""
    T = 16
    x = np.stack([S.make_signal("noise", c, T * hop) for c in range(4)])
    rows = np.stack([np.full(T, 1.5, np.float32), np.full(T, 0.8, np.float32)])
    pv = _pv(fft_size=fft, hop_size=hop, max_channels=4, max_hops=T)
    y = pv.process_batch(x, rows, channels_per_stream=2)
    pv.close()
    for s in range(2):
        o = oracle_lib.Oracle(fft, hop, 2).process_planar(x[2 * s:2 * s + 2], rows[s])
        assert S.rms(y[2 * s:2 * s + 2].astype(np.float64) - o) < REGRESSION_RMS
    pv = _pv(fft_size=fft, hop_size=hop, max_channels=1, max_hops=T)
    mono = pv.process_batch(x[1:2], rows[0])
    pv.close()
    assert np.array_equal(mono[0], y[1])


def test_errors_match_reference_behaviour():
    import phaze_amd
    for bad in (0, 1, 3, 1000):
        with pytest.raises(ValueError, match="FFT size must be a power of two and bigger than 1"):
            phaze_amd.PhaseVocoder(fft_size=bad, hop_size=1)
    with pytest.raises(phaze_amd.PvError):
        phaze_amd.PhaseVocoder(fft_size=1024, hop_size=300)
    pv = phaze_amd.PhaseVocoder(fft_size=1024, hop_size=256, max_channels=1, max_hops=4)
    with pytest.raises(phaze_amd.PvError):
        pv.process_batch(np.zeros((2, 1024), np.float32), np.ones(4, np.float32))     # capacity
    pv.close()


@pytest.mark.parametrize("fft,hop", [(1024, 256), (2048, 512), (4096, 1024), (8192, 2048)])
def test_generic_kernel_fallback_parity(fft, hop):
    """The LDS-staged generic kernel stays the fallback for the shapes the register-resident kernels cover; keep it honest.
    pv_config.flags = PV_FLAG_GENERIC_KERNEL forces it; the two kernels must agree with the oracle and with each other."""
    T = 20
    x = np.stack([S.make_signal("tonal", c, T * hop, stream=1) for c in range(2)])
    p = (0.6 + 1.2 * np.arange(T) / (T - 1)).astype(np.float32)
    yo = oracle_lib.Oracle(fft, hop, 2).process_planar(x, p)
    outs = {}
    for forced in ("0", "1"):
        pv = _pv(fft_size=fft, hop_size=hop, max_channels=2, max_hops=T, flags=1 if forced == "1" else 0)     # PV_FLAG_GENERIC_KERNEL
        outs[forced] = pv.process_batch(x, p)
        name = pv.info()["kernel_name"]
        pv.close()
        assert (name == "pv_chain_kernel") == (forced == "1"), name
        assert S.rms(outs[forced].astype(np.float64) - yo) < REGRESSION_RMS
    assert S.rms(outs["0"].astype(np.float64) - outs["1"]) < 1e-7


@pytest.mark.parametrize("fft,hop", [(1024, 256), (1024, 128), (2048, 512), (4096, 1024)])
def test_residue_fast_and_general_paths(fft, hop):
    """f < 1 with a strong component just below Nyquist: the last region reads the above-Nyquist residue on every frame.  f is stepped across the
    point where the region end crosses N/2 + 1 + N/8 (fast form: decimation identity on the spectrum; beyond it: the re-run stage structure),
    plus silence (no peak at all) and DC."""
    T = 24
    i = np.arange(T * hop, dtype=np.float64)
    hi = 0.3 * np.sin(2 * np.pi * (0.488 * i)) + 0.05 * np.sin(2 * np.pi * 0.031 * i)       # peak near bin 0.976 * N/2
    x = np.stack([(hi + S.make_signal("noise", 0, T * hop).astype(np.float64) / 256).astype(np.float32),
                  np.zeros(T * hop, np.float32), np.full(T * hop, 0.25, np.float32)])
    for f in (0.99, 0.9, 0.8, 0.76, 0.75, 0.745, 0.74, 0.7, 0.6, 0.5, 0.45, 0.3):
        p = np.full(T, f, np.float32)
        pv = _pv(fft_size=fft, hop_size=hop, max_channels=3, max_hops=T)
        y = pv.process_batch(x, p)
        pv.close()
        yo = oracle_lib.Oracle(fft, hop, 3).process_planar(x, p)
        err = S.rms(y.astype(np.float64) - yo)
        assert np.all(np.isfinite(y)) and err < REGRESSION_RMS, f"{fft}/{hop} f={f}: {err:.3e}"
        assert np.max(np.abs(y[1])) == 0.0, "silence stays silence"


@pytest.mark.parametrize("hop", [128, 256, 512, 1024, 2048])
def test_wave2k_kernel(hop):
    """N = 2048: one wave per frame (pv_wave2k_kernel) for every pitchFactor.  f >= 1 (plain stores), 0.75 <= f < 1 (claim rounds + the fast residue),
    f < 0.75 (the residue rebuilt quarter by quarter whenever the last region reads beyond N/2 + N/8), 0, negative, NaN and Inf, chunking, and
    launches of odd length (hop = 128 slides the accumulator by half a register row: the state is handed over in either of its two layouts)."""
    fft, T, nch = 2048, 36, 2
    x = np.stack([S.make_signal("tonal", c, T * hop, stream=9) for c in range(nch)])
    pv = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
    assert pv.info()["kernel_name"] == "pv_wave2k_kernel"
    wg = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T, flags=4)                  # PV_FLAG_WORKGROUP_KERNEL: the second implementation of the path
    assert wg.info()["kernel_name"] == "pv_wg_kernel"
    pw = np.where(np.arange(T) % 3 == 0, 0.6, 1.3).astype(np.float32)
    assert S.rms(wg.process_batch(x, pw).astype(np.float64) - oracle_lib.Oracle(fft, hop, nch).process_planar(x, pw)) < REGRESSION_RMS
    wg.close()
    ar = np.arange(T)
    for pitch in (np.full(T, 0.75, np.float32), np.full(T, 0.8, np.float32), np.full(T, 1.0, np.float32), np.full(T, 1.5, np.float32),
                  np.full(T, 0.5, np.float32), np.full(T, 0.3, np.float32), np.full(T, 0.62, np.float32), np.full(T, 2.0, np.float32),
                  (0.4 + 1.85 * ar / (T - 1)).astype(np.float32),
                  np.where(ar == 17, 0.7, 1.2).astype(np.float32),
                  np.where(ar % 5 == 0, np.nan, 0.9).astype(np.float32),
                  np.where(ar % 7 == 3, 0.0, np.where(ar % 7 == 5, -0.6, 0.55)).astype(np.float32),
                  np.where(ar % 4 == 1, np.inf, 0.45).astype(np.float32)):
        pv.reset()
        y = pv.process_batch(x, pitch)
        yo = oracle_lib.Oracle(fft, hop, nch).process_planar(x, pitch)
        assert np.all(np.isfinite(y))
        assert S.rms(y.astype(np.float64) - yo) < REGRESSION_RMS, pitch[:6]
    # chunked == unchunked == call-split bit for bit, on both residue forms
    for f in (0.85, 0.55):
        p = np.full(T, f, np.float32)
        ref = None
        for F in (T, 5, 11):
            h = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T, frames_per_chunk=F)
            y = h.process_batch(x, p)
            h.close()
            ref = y if ref is None else ref
            assert np.array_equal(y, ref)
        pv.reset()
        cuts = [0, 11, 24, T]
        parts = [pv.process_batch(x[:, a * hop:b * hop], p[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        assert np.array_equal(np.concatenate(parts, axis=1), ref)
    pv.close()


@pytest.mark.parametrize("fft,hop", [(4096, 512), (4096, 1024), (4096, 2048), (4096, 4096), (8192, 1024), (8192, 2048), (8192, 4096), (8192, 8192)])
def test_wg16_kernel(fft, hop):
    """N = 4096 / 8192: sixteen elements per thread, N/32 threads per frame chain (pv_wg16_kernel).  f >= 1 (plain stores), f < 1 (store / add on pairwise
    frames, atomic-MIN claim rounds across the waves otherwise, + the fast residue), f < 0.75 (the residue rebuilt quarter by quarter), 0, negative, NaN
    and Inf, chunking, call splitting, and the same stream through the eight-element workgroup kernel (PV_FLAG_WORKGROUP_KERNEL) as a second
    implementation of the path."""
    T, nch = (28, 3) if fft == 4096 else (20, 2)
    x = np.stack([S.make_signal("tonal" if c != 1 else "noise", c, T * hop, stream=4) for c in range(nch)])
    pv = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
    assert pv.info()["kernel_name"] == "pv_wg16_kernel"
    wg = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T, flags=4)
    assert wg.info()["kernel_name"] == "pv_wg_kernel"
    ar = np.arange(T)
    for pitch in (np.full(T, 1.25, np.float32), np.full(T, 1.0, np.float32), np.full(T, 2.0, np.float32), np.full(T, 0.8, np.float32),
                  np.full(T, 0.75, np.float32), np.full(T, 0.5, np.float32), np.full(T, 0.3, np.float32),
                  (0.4 + 1.85 * ar / (T - 1)).astype(np.float32),
                  np.where(ar % 5 == 0, np.nan, 0.9).astype(np.float32),
                  np.where(ar % 7 == 3, 0.0, np.where(ar % 7 == 5, -0.6, 0.55)).astype(np.float32),
                  np.where(ar % 4 == 1, np.inf, 0.45).astype(np.float32)):
        pv.reset(); wg.reset()
        y = pv.process_batch(x, pitch)
        yo = oracle_lib.Oracle(fft, hop, nch).process_planar(x, pitch)
        assert np.all(np.isfinite(y))
        assert S.rms(y.astype(np.float64) - yo) < REGRESSION_RMS, pitch[:6]
        assert S.rms(wg.process_batch(x, pitch).astype(np.float64) - yo) < REGRESSION_RMS
    wg.close()
    for f in (1.25, 0.85, 0.55):                                               # chunked == unchunked == call-split == repeated, bit for bit
        p = np.full(T, f, np.float32)
        ref = None
        for F in (T, 5, 11, T):
            h = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T, frames_per_chunk=F)
            y = h.process_batch(x, p)
            h.close()
            ref = y if ref is None else ref
            assert np.array_equal(y, ref)
        pv.reset()
        cuts = [0, 9, T - 8, T]
        parts = [pv.process_batch(x[:, a * hop:b * hop], p[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
        assert np.array_equal(np.concatenate(parts, axis=1), ref)
    pv.close()


@pytest.mark.parametrize("fft,hop", [(1024, 256), (2048, 512), (2048, 128), (4096, 1024), (8192, 2048), (512, 128)])
def test_unaligned_device_buffers(fft, hop):
    """pv_process_batch_device promises nothing about alignment beyond float: input / output rows that start 4, 8 or 12 bytes off a 16-byte
    boundary (odd channel stride, offset base) take the kernels' scalar load / store paths and must give the same stream bit for bit."""
    import torch
    T, nch = 20, 3
    x = np.stack([S.make_signal("tonal", c, T * hop, stream=2) for c in range(nch)])
    pitch = np.where(np.arange(T) % 4 == 1, 0.8, 1.3).astype(np.float32)
    yo = oracle_lib.Oracle(fft, hop, nch).process_planar(x, pitch)
    dev = torch.device("cuda:0")
    pt = torch.from_numpy(pitch).to(dev)
    ref = None
    for off, pad in ((0, 0), (1, 1), (2, 3), (3, 0)):
        stride = T * hop + pad
        xin = torch.zeros(nch * stride + 8, device=dev)
        out = torch.full((nch * stride + 8,), 7.0, device=dev)
        for c in range(nch):
            xin[off + c * stride: off + c * stride + T * hop] = torch.from_numpy(x[c]).to(dev)
        pv = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
        pv.process_batch_device(xin.data_ptr() + 4 * off, out.data_ptr() + 4 * off, nch, T, stride, pt.data_ptr())
        pv.synchronize()
        pv.close()
        o = out.cpu().numpy()
        y = np.stack([o[off + c * stride: off + c * stride + T * hop] for c in range(nch)])
        assert S.rms(y.astype(np.float64) - yo) < REGRESSION_RMS, (off, pad)
        if pad:                                                               # nothing written between the rows
            assert np.all(o[off + T * hop: off + stride] == 7.0)
        ref = y if ref is None else ref
        assert np.array_equal(y, ref), (off, pad)


@pytest.mark.parametrize("fft,hop", [(2048, 512), (4096, 1024), (8192, 2048)])
def test_per_stream_pitch_rows(fft, hop):
    """C4's form: several streams of a few channels each in one launch, one pitchFactor row per stream (channel c reads row c // channels_per_stream)."""
    T, nstreams, cps = 18, 3, 2
    nch = nstreams * cps
    x = np.stack([S.make_signal("tonal" if c % 2 else "noise", c, T * hop, stream=c // cps) for c in range(nch)])
    pitch = np.stack([np.full(T, 1.25, np.float32), (0.6 + 1.2 * np.arange(T) / (T - 1)).astype(np.float32), np.where(np.arange(T) % 3 == 0, 0.8, 1.0).astype(np.float32)])
    pv = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
    y = pv.process_batch(x, pitch, channels_per_stream=cps)
    pv.close()
    for s in range(nstreams):
        yo = oracle_lib.Oracle(fft, hop, cps).process_planar(x[s * cps:(s + 1) * cps], pitch[s])
        assert S.rms(y[s * cps:(s + 1) * cps].astype(np.float64) - yo) < REGRESSION_RMS, s
