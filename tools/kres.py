#!/usr/bin/env python
"""Resource table of the kernels of one source file (CPU-only: hipcc cross-compiles for gfx950): VGPRs / AGPRs / SGPRs / spills / scratch / occupancy / code bytes.
    python tools/kres.py pv_wave_kernel.hip [extra hipcc flags ...]
Design aid, not part of the product."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(src, extra=()):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only", "-c",
                          "-Rpass-analysis=kernel-resource-usage", "-o", os.devnull, src, *extra],
                         cwd=os.path.join(ROOT, "phaze_amd", "csrc"), capture_output=True, text=True)
    if out.returncode:
        sys.exit(out.stderr[-4000:])
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
    return kernels


def short(name):
    out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    return re.sub(r"\(anonymous namespace\)::|\(PvKernelParams.*", "", out).replace("void ", "")


if __name__ == "__main__":
    ks = resources(sys.argv[1], sys.argv[2:])
    for k in sorted(ks, key=short):
        v = ks[k]
        print(f"{short(k):62s} VGPR {v.get('VGPRs', 0):3d} AGPR {v.get('AGPRs', 0):3d} SGPR {v.get('TotalSGPRs', v.get('SGPRs', 0)):3d} spillV {v.get('VGPRs Spill', 0):3d} "
              f"spillS {v.get('SGPRs Spill', 0):3d} scratch {v.get('ScratchSize', 0):4d} occ {v.get('Occupancy', 0)} lds {v.get('LDS Size', 0)}")
