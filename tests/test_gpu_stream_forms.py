"""The streaming quantum (pv_process) in its hand-over forms, against each other and against the oracle, over many quanta.  GPU box.

Default on a large-BAR device: the host writes a small quantum into DEVICE memory through the BAR and launches; PV_FLAG_STREAM_PINNED_INPUT: the
kernel reads the hop from pinned host memory; PV_FLAG_PERSISTENT_STREAM: no launch, a resident wave picks the quantum up (control word + input
through the BAR, or through pinned memory with PV_FLAG_STREAM_PINNED_INPUT).  All forms run the same kernel code on the same state: their outputs
must be BIT-identical, quantum by quantum -- a stale cache line or a write overtaken on its way to the device would show up here as a differing hop.
"""
import time

import numpy as np
import pytest

import oracle_lib
import signals as S

pytestmark = pytest.mark.gpu
PINNED, RESIDENT, NO_HDP = 16, 32, 64


def _stream(fft, hop, nch, flags, x, pitch, pauses=()):
    import phaze_amd
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1, flags=flags)
    T = len(pitch)
    y = np.empty((nch, T * hop), np.float32)
    for m in range(T):
        if m in pauses:
            # 80 ms: longer than the resident waves' idle time-out (~50 ms), they have all left; 35 ms: inside the window in which the waves
            # leave one by one -- the library must not hand a quantum to a partly departed launch (it stops and restarts the waves instead)
            time.sleep(pauses[m] if isinstance(pauses, dict) else 0.08)
        blk = [np.ascontiguousarray(x[c, m * hop:(m + 1) * hop]) for c in range(nch)]
        outs = [np.zeros(hop, np.float32) for _ in range(nch)]
        assert pv.process([blk], [outs], {"pitchFactor": np.array([pitch[m]], np.float32)}) is True
        for c in range(nch):
            y[c, m * hop:(m + 1) * hop] = outs[c]
    info = pv.info()
    pv.close()
    return y, info


@pytest.mark.parametrize("fft,hop,nch", [(1024, 256, 1), (1024, 256, 2), (1024, 128, 2), (1024, 512, 1), (2048, 128, 2), (4096, 1024, 8), (4096, 512, 2), (8192, 2048, 8), (8192, 1024, 3)])
def test_every_hand_over_form_gives_the_same_bits(fft, hop, nch):
    T = 6000 if fft <= 2048 else 500
    rng = np.random.default_rng(fft + hop + nch)
    x = (rng.standard_normal((nch, T * hop)) * 0.2).astype(np.float32)           # fresh random data every quantum
    pitch = rng.uniform(0.5, 2.0, T).astype(np.float32)
    base, _ = _stream(fft, hop, nch, PINNED, x, pitch)
    # NO_HDP: the device "does not expose" its HDP flush register -> no hand-over through the BAR may happen (the library falls back to pinned memory)
    for flags in (0, RESIDENT, RESIDENT | PINNED, NO_HDP, NO_HDP | RESIDENT):
        pauses = {100: 0.08, 200: 0.035, 300: 0.045, min(2500, T - 50): 0.08, min(2600, T - 40): 0.025} if flags & RESIDENT else {}
        y, _ = _stream(fft, hop, nch, flags, x, pitch, pauses=pauses)
        bad = np.flatnonzero(np.any(y.view(np.uint32) != base.view(np.uint32), axis=0))
        assert bad.size == 0, f"flags={flags}: first differing sample {bad[0]} (hop {bad[0] // hop}) of {bad.size}"
    K = 40
    ref = oracle_lib.Oracle(fft, hop, nch).process_planar(x[:, :K * hop], pitch[:K])
    assert S.rms(base[:, :K * hop].astype(np.float64) - ref) < 2e-7


def test_resident_kernel_survives_other_calls_on_the_handle():
    """batch call, state export / import, reset and a channel-count change between quanta of a resident stream: the waves leave, the other call
    runs, the next quantum brings them back -- same output as the launch form."""
    import phaze_amd
    fft, hop, T = 1024, 256, 60
    x = np.stack([S.make_signal("tonal", c, T * hop) for c in range(2)])
    pitch = np.full(T, 0.9, np.float32)
    outs = {}
    for flags in (0, RESIDENT):
        pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=2, max_hops=8, flags=flags)
        y = np.zeros((2, T * hop), np.float32)
        m = 0
        while m < T:
            if m == 20:                                            # a batch of 8 hops in the middle of the stream
                y[:, m * hop:(m + 8) * hop] = pv.process_batch(x[:, m * hop:(m + 8) * hop], pitch[m:m + 8])
                m += 8
                continue
            if m == 40:                                            # export + import of channel 0 (a no-op hand-over to itself)
                st = pv.export_state(0)
                pv.import_state(0, *st)
            nch = 1 if 45 <= m < 50 else 2                         # channel-count change (ola-processor.js:38-52: state is reset)
            blk = [np.ascontiguousarray(x[c, m * hop:(m + 1) * hop]) for c in range(nch)]
            o = [np.zeros(hop, np.float32) for _ in range(nch)]
            assert pv.process([blk], [o], {"pitchFactor": pitch[m:m + 1]}) is True
            for c in range(nch):
                y[c, m * hop:(m + 1) * hop] = o[c]
            m += 1
        outs[flags] = y
        pv.close()
    assert np.array_equal(outs[0].view(np.uint32), outs[RESIDENT].view(np.uint32))
