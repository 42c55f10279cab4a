#!/bin/bash
# C5 shape on the default kernel of the build vs PV_FLAG_WORKGROUP_KERNEL (pv_wg_kernel), one call: usage tools/bench_c5_ab.sh <outdir>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=gpurun_out/$1; mkdir -p $OUT
python - <<'PY' | tee $OUT/c5_ab.txt
import sys, os, json, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch, phaze_amd, oracle_lib, signals as S
dev = torch.device('cuda', 0)
fft, hop, nch, T = 8192, 2048, 8, 16384
g = torch.Generator(device=dev); g.manual_seed(5)
n = torch.arange(T * hop, device=dev, dtype=torch.float32)[None, :]
c = torch.arange(nch, device=dev, dtype=torch.float32)[:, None]
x = (0.25 * torch.sin(n * (0.0288 + 0.002 * c)) + 0.125 * torch.sin(n * 0.18) + (torch.rand((nch, T * hop), device=dev, generator=g) - 0.5) / 32).contiguous()
y = torch.empty_like(x)
for label, pt in [("f=1.5", torch.full((T,), 1.5, device=dev)), ("f=0.8", torch.full((T,), 0.8, device=dev)), ("f=0.6", torch.full((T,), 0.6, device=dev)),
                  ("sweep 0.5->2.0", (0.5 + 1.5 * (torch.arange(T, device=dev) % 64).float() / 63.0))]:
    pt = pt.float().contiguous()
    res = {}
    for flags in (0, 4):
        pv = phaze_amd.PhaseVocoder(fft_size=fft, hop_size=hop, max_channels=nch, max_hops=1, flags=flags)
        st = torch.cuda.Stream(device=dev); pv.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            for _ in range(3): pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, pt.data_ptr(), 0, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(8): pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, pt.data_ptr(), 0, 1)
            e1.record(st); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 8
        pv.reset(); pv.process_batch_device(x.data_ptr(), y.data_ptr(), nch, T, T * hop, pt.data_ptr(), 0, 1); pv.synchronize()
        K = 10
        ref = oracle_lib.Oracle(fft, hop, 1).process_planar(x[:1, :K * hop].cpu().numpy(), pt[:K].cpu().numpy())
        err = float(np.sqrt(np.mean((y[:1, :K * hop].cpu().numpy().astype(np.float64) - ref) ** 2)))
        res[flags] = (ms, pv.info()["kernel_name"], err)
        pv.close()
    print(f"{label:16s} {res[0][1]}: {res[0][0]:.3f} ms (rms {res[0][2]:.1e})   {res[4][1]}: {res[4][0]:.3f} ms (rms {res[4][2]:.1e})   ratio {res[0][0] / res[4][0]:.3f}")
PY
