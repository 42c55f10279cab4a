"""The fp32-first forward transform of the one-wave kernels (round 5; DESIGN.md section 3g, phaze_amd/csrc/pv_guard.h): N = 1024 and N = 2048 run the forward FFT in
packed fp32 first and take the peak decisions (/root/reference/src/phase-vocoder.js:95-116) on its magnitudes wherever every comparison they rest on lies outside a guard
band around the fp32 transform's error; other frames ("class B") re-run it in fp64.  A frame's class is a function of its own samples, the ORDER in which a chain runs the
two transforms follows a per-chain counter -- so results must not depend on how a stream is cut into chains or calls, whatever the signal does to the counter.  GPU box,
through the C ABI."""
import numpy as np
import pytest

import oracle_lib
import signals as S

pytestmark = pytest.mark.gpu


def _signal(kind, n, seed=0):
    rng = np.random.default_rng(100 + seed)
    i = np.arange(n, dtype=np.float64)
    if kind == "white":
        return rng.uniform(-0.5, 0.5, n).astype(np.float32)
    if kind == "silence":
        return np.zeros(n, np.float32)
    if kind.startswith("tonal"):                                   # two clean partials over a noise floor at -60 / -80 dB: (nearly) every frame is class B, provably or not
        db = float(kind[5:])
        return (0.5 * np.sin(2 * np.pi * i * 0.0123) + 0.3 * np.sin(2 * np.pi * i * 0.0931) + rng.uniform(-1, 1, n) * 10 ** (-db / 20)).astype(np.float32)
    if kind == "switching":                                        # stretches of noise, clean partials and silence: the counter goes up and down inside every chain
        x = np.zeros(n, np.float32)
        seg = 5000
        for k in range(0, n, seg):
            m = min(seg, n - k)
            which = (k // seg) % 4
            if which == 0:
                x[k:k + m] = rng.uniform(-0.5, 0.5, m)
            elif which == 1:
                x[k:k + m] = 0.5 * np.sin(2 * np.pi * i[k:k + m] * 0.0207)
            elif which == 2:
                x[k:k + m] = 0.4 * np.sin(2 * np.pi * i[k:k + m] * 0.031) + rng.uniform(-1, 1, m) * 1e-3
        return x
    raise ValueError(kind)


@pytest.mark.parametrize("fft,hop", [(1024, 256), (1024, 128), (2048, 512), (2048, 128)])
@pytest.mark.parametrize("kind", ["tonal80", "tonal60", "switching", "white"])
def test_the_order_of_the_two_transforms_never_shows(fft, hop, kind):
    """Unchunked = chunked (5 / 37 frames per chain: every chain starts with a fresh counter) = split into calls = hop by hop, bit for bit, for pitch factors on both sides
    of 1 -- on signals that keep the order counter high, low, and moving."""
    import phaze_amd
    T, nch = 150, 2
    x = np.stack([_signal(kind, T * hop, c) for c in range(nch)])
    for pitch in (np.full(T, 1.5, np.float32), np.full(T, 0.8, np.float32), (0.5 + 1.5 * (np.arange(T) % 64) / 63.0).astype(np.float32)):
        ref = None
        for fpc in (0, 5, 37):
            pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T, frames_per_chunk=fpc)
            y = pv.process_batch(x, pitch)
            if ref is None:
                ref = y
                yo = oracle_lib.Oracle(fft, hop, nch).process_planar(x, pitch)
                assert S.rms(y.astype(np.float64) - yo) < 2e-7
            assert np.array_equal(y.view(np.uint32), ref.view(np.uint32)), (fft, hop, kind, fpc)
            pv.close()
        pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
        parts = [pv.process_batch(x[:, a * hop:b * hop], pitch[a:b]) for a, b in ((0, 1), (1, 44), (44, 45), (45, 131), (131, T))]
        pv.close()
        assert np.array_equal(np.concatenate(parts, axis=1).view(np.uint32), ref.view(np.uint32)), (fft, hop, kind, "call split")
        pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1)         # streaming quanta: every launch is a chain of one frame
        out = np.zeros_like(ref[:, :40 * hop])
        for m in range(40):
            o = [np.zeros(hop, np.float32) for _ in range(nch)]
            pv.process([[x[c, m * hop:(m + 1) * hop] for c in range(nch)]], [o], {"pitchFactor": pitch[m:m + 1]})
            for c in range(nch):
                out[c, m * hop:(m + 1) * hop] = o[c]
        pv.close()
        assert np.array_equal(out.view(np.uint32), ref[:, :40 * hop].view(np.uint32)), (fft, hop, kind, "hop by hop")


@pytest.mark.parametrize("fft,hop", [(1024, 256), (2048, 512)])
def test_forward_stats_and_the_fp64_forward_flag(fft, hop):
    """pv_forward_stats counts what the guard band does: white noise hardly ever falls back, digital silence and clean partials always do; PV_FLAG_FP64_FORWARD runs no fp32
    transform at all and gives the round-4 arithmetic, within 4e-8 of the default."""
    import phaze_amd
    T = 400
    p = np.full(T, 1.25, np.float32)
    rates = {}
    for kind in ("white", "silence", "tonal80"):
        x = _signal(kind, T * hop)[None, :]
        pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=1, max_hops=T)
        y = pv.process_batch(x, p)
        frames, fallbacks = pv.forward_stats()
        assert frames >= T and fallbacks <= frames
        rates[kind] = fallbacks / frames
        assert pv.forward_stats(reset=True) == (frames, fallbacks) and pv.forward_stats() == (0, 0)
        pv.close()
        pv64 = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=1, max_hops=T, flags=phaze_amd.FLAG_FP64_FORWARD)
        y64 = pv64.process_batch(x, p)
        assert pv64.forward_stats() == (0, 0)                        # no fp32-first instance ran
        pv64.close()
        yo = oracle_lib.Oracle(fft, hop, 1).process_planar(x, p)
        assert S.rms(y64.astype(np.float64) - yo) < 2e-7 and S.rms(y.astype(np.float64) - y64.astype(np.float64)) < 4e-8, kind
        if kind != "white":
            assert np.array_equal(y.view(np.uint32), y64.view(np.uint32)), kind      # every frame class B: exactly the fp64-forward kernel's frames
    assert rates["white"] < 0.02 and rates["silence"] == 1.0 and rates["tonal80"] > 0.95, rates


@pytest.mark.parametrize("fft,hop", [(1024, 256), (2048, 512)])
@pytest.mark.parametrize("scale", [1e-9, 3e-6, 1e-3, 1e4, 1e9])
def test_guarded_range_of_amplitudes(fft, hop, scale):
    """The guard test squares magnitudes: frames whose largest bin lies outside [1.1e-7, 1e15) (in |X|^2) are class B by definition.  Signals from far below to far above
    that range, and across its ends inside one stream, match the oracle relative to their own level."""
    import phaze_amd
    T = 60
    base = _signal("white", T * hop, 3).astype(np.float64) + 0.3 * np.sin(2 * np.pi * np.arange(T * hop) * 0.01)
    ramp = np.exp(np.linspace(np.log(0.03), np.log(30.0), T * hop))              # three decades inside the stream: crosses an end of the range for the outer scales
    x = (base * ramp * scale).astype(np.float32)[None, :]
    p = np.full(T, 1.5, np.float32)
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=1, max_hops=T)
    y = pv.process_batch(x, p)
    pv.close()
    yo = oracle_lib.Oracle(fft, hop, 1).process_planar(x, p)
    assert np.all(np.isfinite(y))
    assert S.rms(y.astype(np.float64) - yo) < 2e-7 * max(S.rms(yo), 1e-30) / 0.1


@pytest.mark.parametrize("fft,hop", [(1024, 256), (2048, 512)])
@pytest.mark.parametrize("kind", ["tonal80", "switching"])
def test_resident_waves_carry_the_order_counter_without_showing_it(fft, hop, kind):
    """On the resident kernel (PV_FLAG_PERSISTENT_STREAM) a wave's quanta are ONE chain: its order counter lives across them, so a clean tonal stream runs the fp64
    transform first after a few quanta.  The bits are those of the batch call."""
    import phaze_amd
    T, nch = 80, 2
    x = np.stack([_signal(kind, T * hop, 7 + c) for c in range(nch)])
    pitch = np.full(T, 1.5, np.float32)
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
    ref = pv.process_batch(x, pitch)
    pv.close()
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1, flags=phaze_amd.FLAG_PERSISTENT_STREAM)
    out = np.zeros_like(ref)
    for m in range(T):
        o = [np.zeros(hop, np.float32) for _ in range(nch)]
        pv.process([[x[c, m * hop:(m + 1) * hop] for c in range(nch)]], [o], {"pitchFactor": pitch[m:m + 1]})
        for c in range(nch):
            out[c, m * hop:(m + 1) * hop] = o[c]
    frames, fallbacks = pv.forward_stats()
    pv.close()
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32)), (fft, hop, kind)
    assert frames == T * nch and (kind != "tonal80" or fallbacks >= 0.9 * frames), (frames, fallbacks)
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1, flags=phaze_amd.FLAG_PERSISTENT_STREAM | phaze_amd.FLAG_FP64_FORWARD)
    for m in range(4):                                              # the flag reaches the resident instance too: no fp32-first frame is counted
        o = [np.zeros(hop, np.float32) for _ in range(nch)]
        pv.process([[x[c, m * hop:(m + 1) * hop] for c in range(nch)]], [o], {"pitchFactor": pitch[m:m + 1]})
    assert pv.forward_stats() == (0, 0)
    pv.close()
