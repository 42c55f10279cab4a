"""Host-buffer batches in page-locked memory are pipelined (pv_process_batch: H2D of piece k+1 || kernel of piece k || D2H of piece k-1; pieces =
groups of whole streams, or spans of hops).  The pieces are consecutive calls on the carried state / disjoint channel groups of one pass, so the
result must equal the unpipelined call (pageable buffers) BIT FOR BIT, and state must carry across pipelined calls.  GPU box, through the C ABI."""
import numpy as np
import pytest

import oracle_lib
import signals as S

pytestmark = pytest.mark.gpu


def _x(nch, n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64)
    x = np.empty((nch, n), np.float32)
    for c in range(nch):
        x[c] = (0.25 * np.sin(t * (0.011 + 0.0007 * c)) + 0.1 * np.sin(t * (0.13 + 0.001 * c)) + rng.standard_normal(n) * 0.01).astype(np.float32)
    return x


@pytest.mark.parametrize("fft,hop,nch,T,cps,pf", [
    (1024, 256, 1, 16384, 1, 1.5),        # one long mono stream: pieces are spans of hops
    (1024, 256, 1, 16384, 1, 0.8),
    (2048, 512, 2, 4096, 2, 0.8),         # C3's shape: two channels -> spans of hops
    (4096, 1024, 64, 64, 8, None),        # C4's shape: 8 streams x 8 channels, own pitch row per stream -> groups of whole streams
    (8192, 2048, 8, 128, 8, None),        # C5's shape with its sweep: one 8-channel stream -> spans of hops
    (1024, 256, 96, 192, 2, None),        # 48 stereo streams -> groups of streams, ragged split (48 streams over 9 pieces)
])
def test_pinned_batch_is_pipelined_and_bit_identical(fft, hop, nch, T, cps, pf):
    import phaze_amd
    n = T * hop
    x = _x(nch, n, fft + nch)
    nstreams = nch // cps
    if pf is None:
        pitch = np.stack([(0.5 + 1.5 * ((np.arange(T) + 7 * s) % 64) / 63.0).astype(np.float32) for s in range(nstreams)])
    else:
        pitch = np.full(T, pf, np.float32)
    ref_pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
    ref = ref_pv.process_batch(x, pitch, channels_per_stream=cps)                         # pageable: one piece
    ref_pv.close()
    xin, yout = phaze_amd.pinned_empty((nch, n)), phaze_amd.pinned_empty((nch, n))
    xin[:] = x
    yout[:] = np.nan
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
    pv.process_batch(xin, pitch, channels_per_stream=cps, out=yout)
    assert np.array_equal(yout.view(np.uint32), ref.view(np.uint32))
    assert pv.time_cursor == T * hop
    # state carries across pipelined calls: two more half-batches == one pageable call over the same samples
    ref_pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
    ref_pv.process_batch(x, pitch, channels_per_stream=cps)
    ref2 = ref_pv.process_batch(x, pitch, channels_per_stream=cps)
    ref_pv.close()
    T1 = T // 2
    a_in, a_out = phaze_amd.pinned_empty((nch, T1 * hop)), phaze_amd.pinned_empty((nch, T1 * hop))
    b_in, b_out = phaze_amd.pinned_empty((nch, (T - T1) * hop)), phaze_amd.pinned_empty((nch, (T - T1) * hop))
    a_in[:] = x[:, :T1 * hop]
    b_in[:] = x[:, T1 * hop:]
    pa, pb = (pitch[..., :T1], pitch[..., T1:])
    pv.process_batch(a_in, np.ascontiguousarray(pa), channels_per_stream=cps, out=a_out)
    pv.process_batch(b_in, np.ascontiguousarray(pb), channels_per_stream=cps, out=b_out)
    pv.close()
    got = np.concatenate([a_out, b_out], axis=1)
    assert np.array_equal(got.view(np.uint32), ref2.view(np.uint32))
    K = min(T, 24)
    o = oracle_lib.Oracle(fft, hop, cps).process_planar(x[:cps, :K * hop], np.ascontiguousarray(pitch[0, :K] if pitch.ndim == 2 else pitch[:K]))
    assert S.rms(ref[:cps, :K * hop].astype(np.float64) - o) < 2e-7


def test_pinned_batch_keeps_other_slots_and_a_mixed_buffer_pair_takes_the_plain_path():
    """A pipelined call over nch < used_channels slots carries the other slots' state across the flip(s); a mixed pinned / pageable pair of
    buffers takes the unpipelined path with the same bits.  (The roll-back of a pipelined call that fails in a later piece: test_failed_piece_rolls_the_handle_back below.)"""
    import phaze_amd
    fft, hop, T = 1024, 256, 8192
    x = _x(3, T * hop, 5)
    p = np.full(T, 1.2, np.float32)
    ref_pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=3, max_hops=T)
    r1 = ref_pv.process_batch(x, p)
    r2 = ref_pv.process_batch(x[:2], p)
    r3 = ref_pv.process_batch(x, p)
    ref_pv.close()
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=3, max_hops=T)
    xin, yout = phaze_amd.pinned_empty((3, T * hop)), phaze_amd.pinned_empty((3, T * hop))
    xin[:] = x
    pv.process_batch(xin, p, out=yout)
    assert np.array_equal(yout.view(np.uint32), r1.view(np.uint32))
    pv.process_batch(xin[:2], p, out=yout[:2])                     # two of three slots, pipelined by spans of hops: slot 2 keeps its state
    assert np.array_equal(yout[:2].view(np.uint32), r2.view(np.uint32))
    y3 = pv.process_batch(xin, p)                                  # pinned in, pageable out: unpipelined
    assert np.array_equal(y3.view(np.uint32), r3.view(np.uint32))
    pv.close()


@pytest.mark.parametrize("fft,hop,nch,T,cps", [
    (1024, 256, 1, 16384, 1),             # spans of hops: the pieces commit the state ping-pong one by one -> the snapshot taken before the first piece is put back
    (1024, 256, 96, 192, 2),              # groups of whole streams: nothing is committed before the last piece
])
def test_failed_piece_rolls_the_handle_back(fft, hop, nch, T, cps):
    """PV_FLAG_TEST_FAIL_SECOND_PIECE (a test hook of the C ABI, round 6; advisor r05): a pipelined batch reports PV_ERR_DEVICE behind its second piece, with piece 0 already
    through the kernel (hop spans: state committed, both ping-pong halves about to be overwritten).  The call must fail, leave timeCursor and the channel state exactly as they
    were, and the handle must go on as if the call had never been made: the same batch again == a handle that never saw the failure, bit for bit."""
    import phaze_amd
    n = T * hop
    x = _x(nch, n, 77 + nch)
    pitch = np.full(T, 0.9, np.float32)
    ref_pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T)
    r1 = ref_pv.process_batch(x, pitch, channels_per_stream=cps)
    r2 = ref_pv.process_batch(x, pitch, channels_per_stream=cps)
    ref_pv.close()
    xin, yout = phaze_amd.pinned_empty((nch, n)), phaze_amd.pinned_empty((nch, n))
    xin[:] = x
    # one handle WITH the hook: first a clean call through the plain path (pageable output), so that there is state to roll back to
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=T, flags=phaze_amd.FLAG_TEST_FAIL_SECOND_PIECE)
    y1 = pv.process_batch(xin, pitch, channels_per_stream=cps)
    assert np.array_equal(y1.view(np.uint32), r1.view(np.uint32))
    cur0 = pv.time_cursor
    hist0, acc0, _ = pv.export_state(0)
    with pytest.raises(phaze_amd.PvError) as ei:
        pv.process_batch(xin, pitch, channels_per_stream=cps, out=yout)           # pinned in AND out, >= 4 MB: pipelined -> the injected failure
    assert "injected failure" in str(ei.value)
    assert pv.time_cursor == cur0
    hist1, acc1, _ = pv.export_state(0)
    assert np.array_equal(hist0.view(np.uint32), hist1.view(np.uint32)) and np.array_equal(acc0.view(np.uint32), acc1.view(np.uint32))
    y2 = pv.process_batch(xin, pitch, channels_per_stream=cps)                    # (pageable output: the plain path, no hook) continues from the restored state
    pv.close()
    assert np.array_equal(y2.view(np.uint32), r2.view(np.uint32))
