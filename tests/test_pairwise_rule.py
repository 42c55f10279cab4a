"""The rule behind the claim-free f < 1 scatter of the HIP kernels (pv_wave_kernel.hip, "pairwise"), checked on random peak sets.  CPU-only.

Reference semantics (src/phase-vocoder.js:119-173): every peak p_i moves its region of influence by delta_i = round(p_i f) - p_i; regions tile the
bins, a bin belongs to the peak on its left iff it is strictly closer to it.  For f < 1 the shifted regions overlap and the reference adds (`+=`).
The kernels classify a source bin as "falling side" (owned by the peak on its left, the peak bin included) or "rising side" (owned by the peak on its
right) and claim: while, for every pair of neighbouring peaks, ov = delta_i - delta_{i+1} <= floor((p_{i+1} - p_i) / 2), no target receives two
sources of the same kind and none receives more than two -- so "falling-side sources store, then rising-side sources add" reproduces the reference
sum, in the reference's order.  This test states the regions the way the reference does and checks exactly that claim.
"""
import numpy as np
import pytest

H = 513


def regions(peaks):
    """owner[b] of every source bin b in [0, H): the reference's region tiling (pv:132-141)."""
    owner = np.zeros(H, np.int64)
    for b in range(H):
        j = int(np.searchsorted(peaks, b, side="right")) - 1
        if j < 0:
            owner[b] = 0
        elif j == len(peaks) - 1:
            owner[b] = j
        else:
            owner[b] = j if (b - peaks[j] < peaks[j + 1] - b) else j + 1
    return owner


def random_peaks(rng, mean_gap):
    gaps = 3 + rng.geometric(1.0 / max(mean_gap - 2.0, 1.01), size=400) - 1      # local maxima over +-2 bins are at least 3 bins apart
    p = np.cumsum(gaps)
    return p[p < 511]


@pytest.mark.parametrize("f", [0.97, 0.9, 0.8, 0.75, 0.7, 0.6, 0.5, 0.35])
@pytest.mark.parametrize("mean_gap", [3.5, 5, 8, 16, 40])
def test_pairwise_rule_implies_one_store_one_add(f, mean_gap):
    rng = np.random.default_rng(int(f * 1000) * 97 + int(mean_gap * 10))
    accepted = 0
    for _ in range(120):
        peaks = random_peaks(rng, mean_gap)
        if len(peaks) == 0:
            continue
        delta = np.floor(peaks * np.float64(np.float32(f)) + 0.5).astype(np.int64) - peaks     # Math.round(p * f) - p
        owner = regions(peaks)
        src = np.arange(H)
        rising = src < peaks[owner]
        tgt = src + delta[owner]
        valid = (tgt >= 0) & (tgt < H)
        ok = bool(np.all((delta[:-1] - delta[1:]) <= (np.diff(peaks) >> 1)))                  # the kernels' wave-uniform test
        if not ok:
            continue
        accepted += 1
        fall_t, rise_t = tgt[valid & ~rising], tgt[valid & rising]
        assert len(np.unique(fall_t)) == len(fall_t), "two falling-side sources on one target"
        assert len(np.unique(rise_t)) == len(rise_t), "two rising-side sources on one target"
        # and where a falling and a rising source meet, the falling one belongs to the EARLIER region: store-then-add is the reference's order
        both = np.intersect1d(fall_t, rise_t)
        for t in both:
            bf = src[valid & ~rising & (tgt == t)][0]
            br = src[valid & rising & (tgt == t)][0]
            assert owner[bf] + 1 == owner[br]
    if f >= 0.75:
        assert accepted > 100          # the rule holds for (nearly) every frame in the range the fast residue covers


def test_the_rule_cannot_fail_from_two_thirds_up():
    """The kernels skip the wave-uniform test when pitchFactor >= PV_PAIRWISE_SURE = 0.6667f (pv_device_common.h): by exhaustion over every peak position
    and every gap >= 3 (local maxima over +-2 bins are at least 3 bins apart) for N <= 8192, and for the f32 values around the threshold."""
    thr = np.float32(0.6667)
    fs = [thr, np.nextafter(thr, np.float32(1)), np.float32(0.67), np.float32(0.7), np.float32(0.75), np.float32(0.8), np.float32(0.8333333),
          np.float32(0.9), np.float32(0.99), np.nextafter(np.float32(1), np.float32(0))]
    rng = np.random.default_rng(5)
    fs += list(rng.uniform(float(thr), 1.0, 6).astype(np.float32))
    p = np.arange(0, 4097, dtype=np.int64)[:, None]
    gap = np.arange(3, 4097, dtype=np.int64)[None, :]
    for f in fs:
        assert f >= thr
        fd = np.float64(f)
        d0 = np.floor(p * fd + 0.5).astype(np.int64) - p                          # Math.round(p f) - p
        d1 = np.floor((p + gap) * fd + 0.5).astype(np.int64) - (p + gap)
        assert bool(np.all(d0 - d1 <= (gap >> 1))), f
        assert bool(np.all(np.floor(p * fd + 0.5) <= 4097)) and bool(np.all(np.floor(p * fd + 0.5) >= 0))   # no peak is dropped (pv:127-129)
    # and just below 2/3 it does fail (gap 3): the threshold is not slack by more than the rounding of the constant
    f = np.float64(np.float32(0.6666))
    pp = np.arange(0, 4097, dtype=np.int64)
    assert bool(np.any((np.floor(pp * f + 0.5) - pp) - (np.floor((pp + 3) * f + 0.5) - (pp + 3)) > 1))
    # the sentinels of a missing neighbour (no peak on one side) pass as well: shift 0 at distance >= 4096
    for f in fs:
        fd = np.float64(f)
        sh = (np.floor(pp * fd + 0.5) - pp).astype(np.int64)
        assert bool(np.all(0 - sh <= ((pp + 4096) >> 1))) and bool(np.all(sh - 0 <= ((8192 - pp) >> 1)))
