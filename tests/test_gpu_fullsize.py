"""Parity at BASELINE.json's FULL sizes through size-independent properties (the oracle cannot run these sizes in seconds).

Everything stays on the device (torch tensors + pv_process_batch_device); only small slices come back to the host.
  * K1 identity: pitchFactor = 1  =>  y[n] = 0.375 * x[n - (N - hop)]   (any size, any hop with R = 4)
  * replication: streams fed identical input produce bit-identical output (channel independence, K5)
  * chunk invariance: different frames_per_chunk => bit-identical output
  * prefix parity: the first hops of the full-size run equal the oracle on the same prefix
"""
import numpy as np
import pytest

import oracle_lib
import signals as S

pytestmark = pytest.mark.gpu


def _setup(fft, hop, nch, T, fpc=0):
    import torch
    import phaze_amd
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1, frames_per_chunk=fpc)
    st = torch.cuda.Stream()
    pv.set_stream(st.cuda_stream)
    return torch, pv, st


def _noise(torch, nch, n, seed=0):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return (torch.rand((nch, n), device="cuda", generator=g) - 0.5)


@pytest.mark.parametrize("fft,hop,nch,T", [(1024, 256, 1, 1 << 20), (2048, 512, 2, 1 << 16), (4096, 1024, 8 * 64, 64), (8192, 2048, 8, 1 << 11)])
def test_identity_pf1_full_size(fft, hop, nch, T):
    torch, pv, st = _setup(fft, hop, nch, T)
    x = _noise(torch, nch, T * hop)
    y = torch.empty_like(x)
    p = torch.ones(T, device="cuda")
    torch.cuda.synchronize()          # inputs are produced on torch's default stream, the library launches on `st`
    with torch.cuda.stream(st):
        pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, p.data_ptr())
    pv.synchronize()
    d = fft - hop
    err = (y[:, d:] - 0.375 * x[:, :-d]).double().pow(2).mean().sqrt().item()
    assert err < 1e-7, err
    assert pv.time_cursor == T * hop
    pv.close()


def test_c4_streams_replicated_and_prefix_parity():
    """C4 shape: 4096/1024, 8 channels x 128 streams (one GPU's share of 1024 streams), 64 hops, pf 1.25."""
    fft, hop, nstreams, cps, T = 4096, 1024, 128, 8, 64
    nch = nstreams * cps
    torch, pv, st = _setup(fft, hop, nch, T)
    base = np.stack([S.make_signal("tonal", c, T * hop) for c in range(cps)])
    x = torch.from_numpy(np.tile(base, (nstreams, 1))).cuda()
    y = torch.empty_like(x)
    p = torch.full((T,), 1.25, device="cuda")
    torch.cuda.synchronize()          # inputs are produced on torch's default stream, the library launches on `st`
    with torch.cuda.stream(st):
        pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, p.data_ptr())
    pv.synchronize()
    y0 = y[:cps]
    for s in (1, 17, nstreams - 1):
        assert torch.equal(y[s * cps:(s + 1) * cps], y0)
    K = 12
    ref = oracle_lib.Oracle(fft, hop, cps).process_planar(base[:, :K * hop], np.full(K, 1.25, np.float32))
    assert S.rms(y0[:, :K * hop].cpu().numpy().astype(np.float64) - ref) < 2e-7
    pv.close()


@pytest.mark.parametrize("fft,hop", [(1024, 256), (2048, 512)])
def test_chunk_invariance_full_size(fft, hop):
    T = 1 << 16
    outs = []
    for fpc in (0, 29, 200):
        torch, pv, st = _setup(fft, hop, 1, T, fpc)
        x = _noise(torch, 1, T * hop, seed=5) * 0.5 + 0.25 * torch.sin(torch.arange(T * hop, device="cuda") * 0.05)
        y = torch.empty_like(x)
        p = torch.full((T,), 1.5 if fft == 1024 else 0.8, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            pv.process_batch_device(x.data_ptr(), y.data_ptr(), 1, T, T * hop, p.data_ptr())
        pv.synchronize()
        outs.append(y)
        pv.close()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_c5_sweep_prefix_parity_and_finite():
    """C5 shape: 8192/2048, 8 channels, pitchFactor swept 0.5 -> 2.0 over 256 hops; prefix vs oracle, everything finite."""
    fft, hop, nch, T = 8192, 2048, 8, 256
    torch, pv, st = _setup(fft, hop, nch, T)
    xs = np.stack([S.make_signal("tonal", c, T * hop) for c in range(nch)])
    pitch = (0.5 + 1.5 * np.arange(T, dtype=np.float64) / (T - 1)).astype(np.float32)
    x, p = torch.from_numpy(xs).cuda(), torch.from_numpy(pitch).cuda()
    y = torch.empty_like(x)
    torch.cuda.synchronize()          # inputs are produced on torch's default stream, the library launches on `st`
    with torch.cuda.stream(st):
        pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, p.data_ptr())
    pv.synchronize()
    assert torch.isfinite(y).all()
    K = 10
    ref = oracle_lib.Oracle(fft, hop, 2).process_planar(xs[:2, :K * hop], pitch[:K])
    assert S.rms(y[:2, :K * hop].cpu().numpy().astype(np.float64) - ref) < 2e-7
    pv.close()


@pytest.mark.parametrize("nstreams,T", [(3000, 24), (70000, 5)])
def test_many_short_streams_1024(nstreams, T):
    """N = 1024: many channels with few hops each (one chain per channel, packed 12 to a workgroup across channels; > 65535 channel slots in
    one launch).  Channels carry one of four base signals: equal inputs give bit-identical outputs, the first four match the oracle."""
    fft, hop, cps = 1024, 256, 4
    torch, pv, st = _setup(fft, hop, nstreams, T)
    base = np.stack([S.make_signal("tonal" if c & 1 else "noise", c, T * hop) for c in range(cps)])
    reps = (nstreams + cps - 1) // cps
    x = torch.from_numpy(np.tile(base, (reps, 1))[:nstreams]).cuda()
    y = torch.empty_like(x)
    p = torch.full((T,), 0.9, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        pv.process_batch_device(x.data_ptr(), y.data_ptr(), nstreams, T, T * hop, p.data_ptr())
    pv.synchronize()
    for s in (1, reps // 2, reps - 2):
        assert torch.equal(y[s * cps:(s + 1) * cps], y[:cps])
    ref = oracle_lib.Oracle(fft, hop, cps).process_planar(base, np.full(T, 0.9, np.float32))
    assert S.rms(y[:cps].cpu().numpy().astype(np.float64) - ref) < 2e-7
    pv.close()


@pytest.mark.parametrize("fft,hop,nch,T,pf", [(1024, 256, 1, 1 << 20, 1.5), (1024, 256, 1, 1 << 20, "sweep"), (2048, 512, 2, 1 << 18, 0.8),
                                                (4096, 1024, 16, 64, 1.25), (8192, 2048, 8, 1 << 14, "sweep")])
def test_windows_anywhere_in_a_full_size_run_match_the_oracle(fft, hop, nch, T, pf):
    """BASELINE's full sizes against the ORACLE, not only on a prefix: the output of hop m is the sum of frames m - R + 1 .. m, a frame sees its own N
    input samples, its pitchFactor and (m mod R) only -- so the oracle started at ANY hop a0 = 0 (mod R) on the same input reproduces the full-size
    run from hop a0 + 2 (R - 1) on.  Windows of 48 hops at random offsets, at the first and last hops, and across the kernels' chunk boundaries."""
    torch, pv, st = _setup(fft, hop, nch, T)
    R = fft // hop
    x = _noise(torch, nch, T * hop, seed=11) * 0.2
    n = torch.arange(T * hop, device="cuda", dtype=torch.float32)
    for c in range(nch):
        x[c] += 0.3 * torch.sin(n * (0.031 * (c + 1))) + 0.2 * torch.sin(n * (0.173 + 0.01 * c)) + 0.1 * torch.sin(n * 1.3)
    if pf == "sweep":
        pitch = (0.5 + 1.5 * (torch.arange(T, device="cuda") % 64).float() / 63.0).float().contiguous()
    else:
        pitch = torch.full((T,), float(pf), device="cuda")
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, pitch.data_ptr())
    pv.synchronize()
    fpc = int(pv.info().get("frames_per_chunk", 0)) or 512
    W = min(48, T)
    rng = np.random.default_rng(fft + nch)
    starts = {0, max(T - W, 0) // R * R}
    if T > 4 * W:
        for k in (1, 2, 5):                                   # windows straddling chunk boundaries
            if k * fpc + W < T:
                starts.add((k * fpc - W // 2) // R * R)
        starts |= {int(s) // R * R for s in rng.integers(0, T - W, 6)}
    chans = sorted({0, nch - 1})
    worst = 0.0
    for a0 in sorted(starts):
        xs = x[chans, a0 * hop:(a0 + W) * hop].cpu().numpy()
        ps = pitch[a0:a0 + W].cpu().numpy()
        ref = oracle_lib.Oracle(fft, hop, len(chans)).process_planar(xs, ps)
        skip = 0 if a0 == 0 else 2 * (R - 1)
        got = y[chans, (a0 + skip) * hop:(a0 + W) * hop].cpu().numpy().astype(np.float64)
        worst = max(worst, S.rms(got - ref[:, skip * hop:]))
    assert worst < 2e-7, worst
    pv.close()
