"""Register budget of pv_wg16_kernel (CPU-only: hipcc cross-compiles for gfx950 and reports the resources it allocated).

The kernel runs two waves per SIMD -- two workgroups per CU at N = 8192, four at N = 4096 -- and that rests on 256 registers in all, AGPRs included.  A non-inlined
callee is compiled for the largest budget and the kernel inherits what it takes: a few AGPRs in the general residue once cost every instance its second wave (everything 1.8x
slower, the f >= 1 frames that never call it included).  This test keeps the cliff from coming back unnoticed; the resident (streaming) instances run one wave per SIMD on purpose."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_wg16_instances_keep_two_waves_per_simd():
    src = os.path.join(ROOT, "phaze_amd", "csrc")
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage",
                          "-o", os.devnull, "pv_wg16_kernel.hip"], cwd=src, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels = {}
    name = None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            kernels[name][m.group(1).strip()] = int(m.group(2))
    inst = {k: v for k, v in kernels.items() if "pv_wg16_kernel" in k}
    assert len(inst) == 24, sorted(inst)                         # 2 sizes x 4 hops x (product, tap, resident)
    for k, v in inst.items():
        log2n, rows, aux, resident = re.search(r"ILi(\d+)ELi(\d+)ELb([01])ELb([01])E", k).groups()
        if resident == "1":
            assert v["Occupancy"] == 1 and v["VGPRs Spill"] == 0, (k, v)      # the whole register file, AGPRs instead of scratch
            continue
        assert v["Occupancy"] == 2 and v["AGPRs"] == 0, (k, v)
        assert v["VGPRs"] <= 256
        if aux == "0":
            assert v["VGPRs Spill"] <= (0 if rows == "4" else 12), (k, v)   # BASELINE's shapes (hop = N/4): nothing in scratch; the other hops a handful around the residue call
        lds = 81632 if log2n == "13" else 39840
        assert (lds + 256 + 511) // 512 * 512 * (2 if log2n == "13" else 4) <= 160 * 1024


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_headline_instances_keep_three_waves_per_simd_and_their_hot_path_out_of_scratch():
    """pv_wave_kernel_1024: every batch instance runs three waves per SIMD (<= 168 VGPRs, no AGPRs).  The fp32-first instances for pitchFactor >= 1 chains (the
    headline launch) carry their rare forward transforms out of line: inlined, the register allocator parked values of the HOT path in scratch memory (round 5:
    62 spilled VGPRs, ~18 scratch accesses per frame).  A handful of loop invariants of the inline fp64-first path may live there; more means the hot path is hit again."""
    src = os.path.join(ROOT, "phaze_amd", "csrc")
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage",
                          "-o", os.devnull, "pv_wave_kernel.hip"], cwd=src, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            kernels[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            kernels[name][m.group(1).strip()] = int(m.group(2))
    inst = {k: v for k, v in kernels.items() if "pv_wave_kernel_1024" in k}
    assert len(inst) == 28, sorted(inst)                         # 4 hops x (2 classes x 2 forward forms + tap + 2 resident)
    for k, v in inst.items():
        rows, aux, resident, spread, f32 = re.search(r"ILi(\d+)ELb([01])ELb([01])ELb([01])ELb([01])E", k).groups()
        if resident == "1":
            continue
        assert v["Occupancy"] == 3 and v["AGPRs"] == 0 and v["VGPRs"] <= 168, (k, v)
        if spread == "1" and f32 == "0":
            assert v["VGPRs Spill"] == 0 and v["ScratchSize"] == 0, (k, v)
        if spread == "1" and f32 == "1":
            assert v["VGPRs Spill"] <= 8, (k, v)
