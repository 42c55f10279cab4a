"""Seeded fuzz of the HIP path against the CPU oracle: random FFT sizes, hops, channel counts, chunk lengths, call splits and pitch
schedules (incl. f < 0.5, f > 2, exact ties, 0, negatives and non-finite values), through every kernel variant.  Deterministic (fixed seeds)."""
import numpy as np
import pytest

import oracle_lib
import signals as S

pytestmark = pytest.mark.gpu


def _case(rng):
    log2n = int(rng.integers(6, 14))
    N = 1 << log2n
    hop = N >> int(rng.integers(0, min(log2n - 1, 5) + 1))
    if N == 8192 and hop < 1024:
        hop = 1024                                   # LDS limit of the generic kernel at 8192 (documented in include/phaze_amd.h)
    nch = int(rng.integers(1, 4))
    T = int(rng.integers(1, 40 if N <= 2048 else 14))
    kind = ["noise", "tonal"][int(rng.integers(0, 2))]
    mode = int(rng.integers(0, 4))
    if mode == 0:
        p = rng.uniform(0.3, 3.0, size=T)
    elif mode == 1:
        p = np.full(T, rng.choice([0.5, 0.75, 1.0, 1.5, 2.0, 0.33, 2.5]))
    elif mode == 2:
        p = rng.uniform(0.05, 0.6, size=T)
    else:
        p = rng.choice([0.0, -1.0, 0.8, 1.2, np.nan, np.inf, -np.inf, 100.0, 1e-3], size=T)
    return N, hop, nch, T, kind, p.astype(np.float32)


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_against_oracle(seed):
    import phaze_amd
    rng = np.random.default_rng(1234 + seed)
    worst = 0.0
    for _ in range(14):
        N, hop, nch, T, kind, p = _case(rng)
        x = np.stack([S.make_signal(kind, c, T * hop, stream=seed) for c in range(nch)])
        fpc = int(rng.choice([0, 0, 1, 3, 7, 16]))
        pv = phaze_amd.PhaseVocoder(fft_size=N, hop_size=hop, max_channels=nch, max_hops=T, frames_per_chunk=fpc)
        parts, pos = [], 0
        while pos < T:                               # random call splitting: state must carry exactly
            n = int(rng.integers(1, T - pos + 1))
            parts.append(pv.process_batch(x[:, pos * hop:(pos + n) * hop], p[pos:pos + n]))
            pos += n
        y = np.concatenate(parts, axis=1)
        name = pv.info()["kernel_name"]
        pv.close()
        yo = oracle_lib.Oracle(N, hop, nch).process_planar(x, p)
        assert np.all(np.isfinite(y)), (N, hop, kind, name)
        err = S.rms(y.astype(np.float64) - yo)
        worst = max(worst, err)
        assert err < 2e-7, f"N={N} hop={hop} nch={nch} T={T} {kind} fpc={fpc} kernel={name}: rms {err:.3e}"
    print("seed", seed, "worst rms", worst)


@pytest.mark.parametrize("fft,hop,pf", [(1024, 256, 1.5), (1024, 256, 0.8), (2048, 128, 1.0), (2048, 512, 0.8), (4096, 1024, 1.25), (8192, 2048, 0.9),
                                        (256, 64, 1.3), (1024, 128, 1.5)])
@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_non_finite_samples_poison_the_same_hops_as_the_reference(fft, hop, pf, bad):
    """A NaN / Inf sample in the input: in the reference every comparison with a NaN magnitude fails to reject (pv:103,107), peaks appear at
    every other bin, and the NaNs they carry reach every output sample of the frames whose window holds the sample.  The register kernels
    compare magnitudes as bit patterns, which is only the float order for finite values: they detect a non-finite magnitude per frame and emit
    the frame as NaN.  Hops outside the reach of the bad sample must be finite and equal to the oracle's; the NaN hops must be the same hops."""
    import phaze_amd
    T, nch = 24, 2
    x = np.stack([S.make_signal("tonal", c, T * hop) for c in range(nch)])
    x[1, 9 * hop + hop // 3] = bad                              # channel 1 only: channel 0 must stay clean (channels are independent, K5)
    p = np.full(T, pf, np.float32)
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
    y = pv.process_batch(x, p)
    name = pv.info()["kernel_name"]
    pv.close()
    with np.errstate(all="ignore"):
        yo = oracle_lib.Oracle(fft, hop, nch).process_planar(x, p)
    bad_ref = ~np.isfinite(yo)
    bad_got = ~np.isfinite(y)
    assert not bad_got[0].any() and not bad_ref[0].any(), name
    hops_ref = bad_ref[1].reshape(T, hop).any(axis=1)
    hops_got = bad_got[1].reshape(T, hop).any(axis=1)
    assert hops_ref.any() and np.array_equal(hops_ref, hops_got), (name, np.nonzero(hops_ref)[0], np.nonzero(hops_got)[0])
    # inside a poisoned hop the reference's samples are all non-finite; so are ours
    assert bad_got[1].reshape(T, hop)[hops_got].all() and bad_ref[1].reshape(T, hop)[hops_ref].all(), name
    ok = ~bad_ref
    assert S.rms((y.astype(np.float64) - yo)[ok]) < 2e-7, name
