"""AddressSanitizer + UBSan over the host-side C of this repo: the CPU restatement (oracle), the plain-C consumer of the C ABI
(examples/pv_stream.c) and the N-API addon (phaze_amd/node/phaze_napi.c, loaded into Node with libasan preloaded).  CPU: the restatement in
full, the consumers up to the loud "no device" failure.  GPU: the consumers through complete runs (quanta, batches, the asynchronous batch,
state export / import).  The HIP library itself is built by hipcc without sanitizers; its host side is exercised through these callers."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
pytestmark = pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")


def _clean(r):
    txt = r.stdout + r.stderr
    assert "AddressSanitizer" not in txt and "runtime error:" not in txt, txt[-3000:]


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _libdir():
    import phaze_amd
    if not os.path.exists(phaze_amd.library_path()):
        phaze_amd.build_library()
    return os.path.dirname(phaze_amd.library_path())


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "build", "oracle_asan")], capture_output=True, text=True, timeout=600,
                       env=dict(ENV, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stderr[-3000:]
    _clean(r)
    assert "oracle sanitizer driver ok" in r.stdout


def _build_example(tmp_path):
    libdir = _libdir()
    exe = str(tmp_path / "pv_stream_asan")
    cmd = ["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-Wall", "-Wextra"] + SAN + ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "pv_stream.c"),
           "-o", exe, "-L", libdir, "-lphaze_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_consumer_error_path_under_asan(tmp_path):
    exe = _build_example(tmp_path)
    if not _has_gpu():
        r = subprocess.run([exe], capture_output=True, text=True, env=ENV, timeout=300)
        assert r.returncode != 0 and "HIP device error" in r.stderr
        _clean(r)


@pytest.mark.gpu
def test_c_consumer_full_run_under_asan(tmp_path):
    exe = _build_example(tmp_path)
    r = subprocess.run([exe, "1024", "256", "0.8", "48"], capture_output=True, text=True, env=ENV, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    _clean(r)
    assert json.loads(r.stdout.strip().splitlines()[-1])["stream_equals_batch"] is True


NODE = shutil.which("node")
ADDON_JS = r"""
const path = require('path');
const m = require(process.argv[1]);
const out = {exports: Object.keys(m).length};
try { m.create({fftSize: 1000, hopSize: 250}); out.bad = 'no throw'; } catch (e) { out.bad = e.message; }
try { m.process({}, [], [], 1); out.nohandle = 'no throw'; } catch (e) { out.nohandle = 'throws'; }
out.devices = m.deviceCount();
(async () => {
  if (out.devices > 0) {
    const h = m.create({fftSize: 1024, hopSize: 256, maxChannels: 2, maxHops: 8});
    const x = new Float32Array(2 * 8 * 256).map((_, i) => 0.3 * Math.sin(i * 0.05) + 0.01 * Math.sin(i * 1.7));
    const y = new Float32Array(x.length), z = new Float32Array(x.length), p = new Float32Array(8).fill(0.8);
    m.processBatch(h, x, y, 2, 8, p, 0, 1);
    m.reset(h);
    const pr = m.processBatchAsync(h, x, z, 2, 8, p, 0, 1);
    try { m.info(h); out.busy = 'no throw'; } catch (e) { out.busy = e.code; }
    await pr;
    out.async_equal = Buffer.compare(Buffer.from(y.buffer), Buffer.from(z.buffer)) === 0;
    const st = m.exportState(h, 1);
    m.importState(h, 0, st.hist, st.acc, st.timeCursor);
    const o = [new Float32Array(256), new Float32Array(256)];
    m.processBegin(h, [x.subarray(0, 256), x.subarray(256, 512)], 1.5);
    m.processEnd(h, o, 2);
    out.quantum = o[0].some((v) => v !== 0);
    // round 4: fewer output arrays than begun channels (ADVICE r3: the addon passes exactly the begun count, NULL for the missing ones)
    m.processBegin(h, [x.subarray(0, 256), x.subarray(256, 512)], 1.5);
    m.processEnd(h, [o[0]]);
    // pinned buffers: pipelined batch through external ArrayBuffers, strided rows, window of the worker thread
    const px = m.allocPinned(2 * 8 * 256 + 512), py = m.allocPinned(2 * 8 * 256 + 512);
    px.set(x.subarray(0, 8 * 256), 0); px.set(x.subarray(8 * 256), 8 * 256 + 256);
    m.reset(h);
    await m.processBatchAsync(h, px, py, 2, 8, p, 0, 1, 8 * 256 + 256);
    out.pinned_equal = Buffer.compare(Buffer.from(py.buffer, 0, 8 * 256 * 4), Buffer.from(y.buffer, 0, 8 * 256 * 4)) === 0
                    && Buffer.compare(Buffer.from(py.buffer, (8 * 256 + 256) * 4, 8 * 256 * 4), Buffer.from(y.buffer, 8 * 256 * 4, 8 * 256 * 4)) === 0;
    const w = m.batchWindow(h);
    out.window_ok = w.length === 2 && w[1] >= w[0] && w[0] > 0;
    m.destroy(h);
    try { m.info(h); out.destroyed = 'no throw'; } catch (e) { out.destroyed = e.code; }
  } else {
    try { m.create({fftSize: 1024, hopSize: 256}); out.nodev = 'no throw'; } catch (e) { out.nodev = e.message; }
  }
  console.log(JSON.stringify(out));
})().catch((e) => { console.error(e); process.exit(1); });
"""


def _build_addon(tmp_path):
    libdir = _libdir()
    out = str(tmp_path / "phaze_napi_asan.node")
    cmd = ["gcc", "-std=c11", "-fPIC", "-shared", "-Wall", "-Wextra", "-Wno-unused-parameter"] + SAN + ["-DNODE_GYP_MODULE_NAME=phaze_napi", "-I/usr/include/node",
           "-o", out, os.path.join(ROOT, "phaze_amd", "node", "phaze_napi.c"), "-L", libdir, "-lphaze_amd", "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def _run_addon(tmp_path):
    addon = _build_addon(tmp_path)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    r = subprocess.run([NODE, "-e", ADDON_JS, addon], capture_output=True, text=True, timeout=600, env=dict(ENV, LD_PRELOAD=asan), cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    _clean(r)
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_napi_addon_error_paths_under_asan(tmp_path):
    out = _run_addon(tmp_path)
    assert out["exports"] == 16 and out["bad"] == "FFT size must be a power of two and bigger than 1" and out["nohandle"] == "throws"
    if out["devices"] == 0:
        assert "no HIP device" in out["nodev"]


@pytest.mark.gpu
@pytest.mark.skipif(NODE is None, reason="node not installed")
def test_napi_addon_full_surface_under_asan(tmp_path):
    out = _run_addon(tmp_path)
    assert out["devices"] >= 1 and out["busy"] == "PV_BUSY" and out["async_equal"] is True and out["quantum"] is True and out["destroyed"] == "PV_6"
    assert out["pinned_equal"] is True and out["window_ok"] is True
