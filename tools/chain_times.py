#!/usr/bin/env python
"""How unevenly the chains of the headline launch finish (profiles/r05_chain_times.md): per-chain shader-clock time from the stamps build of pv_wave_kernel_1024.
    make -C phaze_amd/csrc variant NAME=stamps FILE=pv_wave_kernel EXTRA=-DPV_STAMPS=1 CAPI_EXTRA=-DPV_STAMPS=1
    PHAZE_LIB=build/exp/libphaze_stamps.so python tools/chain_times.py
Design aid (round 5), not product."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, phaze_amd, bench
fft, hop, T = 1024, 256, 1 << 20
dev = torch.device("cuda", 0)
import json
res = {}
for sig in ("bench", "white"):
    x = bench.synth_input(torch, 1, T * hop, dev, 0) if sig == "bench" else (torch.rand((1, T * hop), device=dev) - 0.5)
    y = torch.empty_like(x)
    p = torch.full((T,), 1.5, device=dev, dtype=torch.float32)
    pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=1, max_hops=1)
    L = pv._L
    for _ in range(3):
        pv.process_batch_device(x.data_ptr(), y.data_ptr(), 1, T, T * hop, p.data_ptr(), 0, 1)
    pv.synchronize()
    info = pv.info()
    nchains = (T + info["frames_per_chunk"] - 1) // info["frames_per_chunk"]
    buf = np.zeros((nchains, 16), np.uint32)
    L.pv_exp_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert L.pv_exp_read_stamps(pv._h, buf.ctypes.data_as(C.c_void_p), nchains) == 0
    fr, fb = pv.forward_stats()
    pv.close()
    tot = buf[:, 13].astype(np.float64)       # ticks of the whole chain (3 launches accumulated)
    res[sig] = {"max_over_mean": float(tot.max() / tot.mean()), "p99_over_mean": float(np.percentile(tot, 99) / tot.mean()), "fallback_rate": fb / max(fr, 1), "chains": int(nchains)}
    print(sig, "chains", nchains, "fpc", info["frames_per_chunk"], "fallback", fb / max(fr, 1), "chain ticks mean", tot.mean(), "max", tot.max(), "max/mean", tot.max() / tot.mean(),
          "p99/mean", np.percentile(tot, 99) / tot.mean(), "min/mean", tot.min() / tot.mean())
    # per workgroup of 12 chains (one CU): the CU is busy until its slowest chain ends
    k = (nchains // 12) * 12
    wg = tot[:k].reshape(-1, 12)
    print("   max of workgroup means / mean", wg.mean(1).max() / tot.mean(), "std of chain / mean", tot[tot > 0.5 * tot.mean()].std() / tot.mean(), "std of wg mean / mean", wg.mean(1).std() / tot.mean())
    print("   per workgroup: mean of max", wg.max(1).mean() / tot.mean(), "max of max", wg.max(1).max() / tot.mean(), "mean of mean", wg.mean(1).mean() / tot.mean())

if len(sys.argv) > 1:      # profiles/chain_tail.json: what bench.py reports as tail_over_mean while the kernels' hash matches
    ent = dict(res["bench"], white_noise=res["white"], csrc_sha16=bench.csrc_sha16(),
               source="tools/chain_times.py on the stamps build (-DPV_STAMPS=1): slowest chain of the headline launch over the mean chain, shader-clock ticks")
    json.dump({"1024/256/ch1/hops1048576": ent}, open(sys.argv[1], "w"), indent=1)
