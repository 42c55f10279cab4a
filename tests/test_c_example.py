"""examples/pv_stream.c: the C ABI consumed from plain C99 (no Python, no Node).  CPU: the header is valid pedantic C99, the program links against
the shared library and fails loudly without a GPU.  GPU: quantum-by-quantum pv_process equals one pv_process_batch bit for bit."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    import phaze_amd
    if not os.path.exists(phaze_amd.library_path()):
        phaze_amd.build_library()
    libdir = os.path.dirname(phaze_amd.library_path())
    exe = str(tmp_path / "pv_stream")
    cmd = ["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-O2", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "pv_stream.c"), "-o", exe, "-L", libdir, "-lphaze_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
           "-L/opt/rocm/lib", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_example_builds_as_pedantic_c99_and_fails_loudly_without_a_gpu(tmp_path):
    exe = _build(tmp_path)
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode != 0 and "HIP device error" in r.stderr                    # no CPU fallback behind the C ABI


@pytest.mark.gpu
@pytest.mark.parametrize("fft,hop,pitch", [(2048, 128, 1.5), (1024, 256, 0.8), (4096, 1024, 1.25), (8192, 2048, 0.6)])
def test_c_example_stream_equals_batch(tmp_path, fft, hop, pitch):
    exe = _build(tmp_path)
    r = subprocess.run([exe, str(fft), str(hop), str(pitch), "96"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["stream_equals_batch"] is True and j["output_rms"] > 1e-3
