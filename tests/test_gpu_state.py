"""State export / import (SURVEY section 5 "checkpoint / resume", 8e "split the time axis"), timeCursor validation and run-to-run
reproducibility of the colliding (f < 1) scatter on every kernel -- all through the C ABI on the GPU."""
import numpy as np
import pytest

import oracle_lib
import signals as S

pytestmark = pytest.mark.gpu
REGRESSION_RMS = 2e-7
GENERIC = 1          # PV_FLAG_GENERIC_KERNEL


def _pv(**kw):
    import phaze_amd
    return phaze_amd.PhaseVocoder(**kw)


@pytest.mark.parametrize("fft,hop,pf,flags", [(1024, 256, 1.5, 0), (1024, 256, 0.7, 0), (2048, 512, 0.8, 0), (2048, 128, 1.2, 0), (512, 128, 0.6, 0),
                                              (1024, 256, 0.7, GENERIC)])
def test_export_import_hands_a_stream_to_another_handle(fft, hop, pf, flags):
    """One C2-like stream split into two spans on TWO handles: the second imports what the first exported (N - hop input samples,
    N - hop pending overlap-add sums, timeCursor: ola-processor.js:59,77, phase-vocoder.js:31) and continues bit for bit."""
    T, T1, nch = 48, 19, 2
    x = np.stack([S.make_signal("tonal", c, T * hop) for c in range(nch)])
    p = np.full(T, pf, np.float32)
    one = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T, flags=flags)
    ref = one.process_batch(x, p)
    one.close()
    a = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T, flags=flags)
    b = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T, flags=flags)
    ya = a.process_batch(x[:, :T1 * hop], p[:T1])
    for c in range(nch):
        hist, acc, tc = a.export_state(c)
        assert tc == T1 * hop and hist.shape == (fft - hop,)
        assert np.array_equal(hist, x[c, T1 * hop - (fft - hop):T1 * hop])        # the history IS the input tail (ola:121-127)
        b.import_state(c, hist, acc, tc)
    assert b.time_cursor == T1 * hop
    yb = b.process_batch(x[:, T1 * hop:], p[T1:])
    a.close(); b.close()
    assert np.array_equal(np.concatenate([ya, yb], axis=1), ref)


@pytest.mark.parametrize("fft,hop,pf", [(1024, 256, 1.5), (2048, 512, 0.8), (2048, 128, 1.5)])
def test_time_sharding_without_hand_over(fft, hop, pf):
    """A span of one stream needs nothing from its predecessor: import {input tail, acc = 0, cursor} R - 1 hops early, recompute the halo
    frames (their output is discarded), and the rest of the span equals the single-handle run bit for bit.  This is how a single mono or
    stereo stream is spread over several GPUs (bench.py --time-shard)."""
    T, nch, R = 60, 2, fft // hop
    L = fft - hop
    x = np.stack([S.make_signal("noise", c, T * hop) for c in range(nch)])
    p = np.full(T, pf, np.float32)
    one = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
    ref = one.process_batch(x, p)
    one.close()
    spans = [(0, 17), (17, 41), (41, T)]
    out = []
    for lo, hi in spans:
        h = _pv(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
        start = max(lo - (R - 1), 0)
        if start > 0:
            for c in range(nch):
                hist = np.zeros(L, np.float32)                            # a stream younger than N - hop samples still has leading zeros (ola:59)
                n0 = min(L, start * hop)
                hist[L - n0:] = x[c, start * hop - n0:start * hop]
                h.import_state(c, hist=hist, acc=np.zeros(L, np.float32), time_cursor=start * hop)
        y = h.process_batch(x[:, start * hop:hi * hop], p[start:hi])
        out.append(y[:, (lo - start) * hop:])
        h.close()
    assert np.array_equal(np.concatenate(out, axis=1), ref)


def test_time_cursor_must_be_a_multiple_of_hop():
    """the register kernels rely on t = m * hop (pv:71 only ever adds hopSize): other values are rejected, multiples work and
    agree with the oracle started at the same cursor."""
    import phaze_amd
    fft, hop, T = 1024, 256, 16
    pv = _pv(fft_size=fft, hop_size=hop, max_channels=1, max_hops=T)
    with pytest.raises(phaze_amd.PvError):
        pv.time_cursor = 100
    with pytest.raises(phaze_amd.PvError):
        pv.time_cursor = -256
    with pytest.raises(phaze_amd.PvError):
        pv.import_state(0, None, None, 257)
    k = 7
    pv.time_cursor = k * hop
    x = S.make_signal("tonal", 0, (k + T) * hop)
    p = np.full(k + T, 1.5, np.float32)
    # the oracle reaches the same cursor by consuming k hops of silence first: state stays zero, only timeCursor moves
    o = oracle_lib.Oracle(fft, hop, 1)
    o.process_planar(np.zeros((1, k * hop), np.float32), p[:k])
    yo = o.process_planar(x[None, k * hop:], p[k:])
    y = pv.process_batch(x[None, k * hop:], p[k:])
    pv.close()
    assert S.rms(y.astype(np.float64) - yo) < REGRESSION_RMS


@pytest.mark.parametrize("fft,hop,flags", [(512, 128, 0), (1024, 256, GENERIC), (1024, 256, 0), (2048, 512, 0), (4096, 1024, 0), (2048, 512, GENERIC)])
@pytest.mark.parametrize("pf", [0.3, 0.45, 0.0])
def test_colliding_scatter_is_reproducible_for_every_f(fft, hop, flags, pf):
    """f < 0.5 piles three and more regions onto one target bin (pv:169-170 `+=`).  The claim rounds add them in ascending source order
    (atomic MIN on the claim word) on the multi-wave kernels, so chunked / unchunked / call-split / repeated runs agree bit for bit;
    round 1's generic kernel used float atomics and did not."""
    T = 24
    x = np.stack([S.make_signal("tonal", c, T * hop, stream=5) for c in range(2)])
    p = np.full(T, pf, np.float32)
    ref = None
    for F in (T, 5, 9, T):
        pv = _pv(fft_size=fft, hop_size=hop, max_channels=2, max_hops=T, frames_per_chunk=F, flags=flags)
        y = pv.process_batch(x, p)
        if F == T and ref is not None:
            y2 = pv.process_batch(x, p)          # state carried on: different, but must be reproducible below
            del y2
        pv.close()
        if ref is None:
            ref = y
        else:
            assert np.array_equal(y, ref), f"frames_per_chunk={F}"
    pv = _pv(fft_size=fft, hop_size=hop, max_channels=2, max_hops=T, frames_per_chunk=4, flags=flags)
    parts, pos = [], 0
    for n in (2, 7, 1, 14):
        parts.append(pv.process_batch(x[:, pos * hop:(pos + n) * hop], p[pos:pos + n]))
        pos += n
    pv.close()
    assert np.array_equal(np.concatenate(parts, axis=1), ref)
    yo = oracle_lib.Oracle(fft, hop, 2).process_planar(x, p)
    assert S.rms(ref.astype(np.float64) - yo) < REGRESSION_RMS


def test_failed_call_leaves_the_handle_as_it_was():
    """A synchronous entry point commits its state (ping-pong half, timeCursor) only when the whole call succeeded.  A launch the library must
    refuse -- more than 65535 channel slots on a kernel that maps channels to grid.y -- returns an error, and the handle continues exactly like
    one that never saw the call."""
    import phaze_amd
    fft, hop, T = 256, 64, 12                      # pv_chain_kernel (not a wave kernel): channels sit on grid.y
    big = 65540
    x = np.stack([S.make_signal("tonal", c, 2 * T * hop) for c in range(2)])
    p = np.full(T, 1.3, np.float32)
    pv = _pv(fft_size=fft, hop_size=hop, max_channels=big, max_hops=T)
    ref = _pv(fft_size=fft, hop_size=hop, max_channels=2, max_hops=T)
    y0 = pv.process_batch(x[:, :T * hop], p)
    r0 = ref.process_batch(x[:, :T * hop], p)
    assert np.array_equal(y0, r0)
    cursor = pv.time_cursor
    xb = np.zeros((big, T * hop), np.float32)
    with pytest.raises(phaze_amd.PvError) as ei:
        pv.process_batch(xb, p)
    assert ei.value.status == phaze_amd.capi.PV_ERR_CAPACITY
    assert pv.time_cursor == cursor == T * hop
    y1 = pv.process_batch(x[:, T * hop:], p)
    r1 = ref.process_batch(x[:, T * hop:], p)
    pv.close(); ref.close()
    assert np.array_equal(y1, r1)


@pytest.mark.parametrize("fft,hop", [(1024, 256), (2048, 512), (4096, 1024), (8192, 2048), (512, 128)])
def test_debug_frame_leaves_unused_slots_zero(fft, hop):
    """pv_debug_frame runs a frame of slots [0, ch] without committing, but the kernels still store that frame's history / accumulator into the
    other ping-pong half.  Slots that have never been processed count as "zero in both halves" (they are not copied across a flip): after a tap
    on slot 1 of a fresh handle, a mono batch and then a stereo batch, channel 1 must start from silence -- as on a handle that was never tapped."""
    T = 12
    x = np.stack([S.make_signal("tonal", c, 2 * T * hop) for c in range(2)])
    p = np.full(T, 0.9, np.float32)
    outs = []
    for tap in (False, True):
        pv = _pv(fft_size=fft, hop_size=hop, max_channels=2, max_hops=T)
        if tap:
            pv.debug_frame(1, x[1, :hop] * 3.0 + 0.5, 1.3)
        y1 = pv.process_batch(x[:1, :T * hop], p)
        y2 = pv.process_batch(x[:, T * hop:], p)
        pv.close()
        outs.append((y1, y2))
    assert np.array_equal(outs[0][0].view(np.uint32), outs[1][0].view(np.uint32))
    assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))
