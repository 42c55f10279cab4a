"""The Node.js host + N-API addon (the drop-in surface of src/phase-vocoder.js).

CPU part (`not gpu`): the addon loads, exports every entry point, the processor class has the reference's
surface and error behaviour.  GPU part: process() driven hop by hop from Node against the golden vectors.
"""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import signals as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "phaze_amd", "node", "phaze_napi.node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node not installed")


def _build():
    if not os.path.exists(ADDON):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "phaze_amd", "csrc")], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "phaze_amd", "node")], stdout=subprocess.DEVNULL)


def _node(script):
    return subprocess.run([NODE, "-e", script], cwd=ROOT, capture_output=True, text=True, timeout=120)


def test_addon_loads_and_exports_surface():
    _build()
    r = _node("""
      const m = require('./phaze_amd/node/phase-vocoder.js');
      const P = m.PhaseVocoderProcessor;
      console.log(JSON.stringify({
        native: Object.keys(m.native).sort(),
        desc: P.parameterDescriptors,
        registered: m.getProcessor('phase-vocoder-processor') === P,
        hasProcess: typeof P.prototype.process === 'function',
      }));""")
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["native"] == sorted(["create", "destroy", "process", "processBatch", "reset", "timeCursor", "info", "processBegin", "processEnd",
                                  "processBatchAsync", "exportState", "importState", "deviceCount", "allocPinned", "batchWindow", "forwardStats"])
    assert d["desc"] == [{"name": "pitchFactor", "defaultValue": 1}]          # phase-vocoder.js:17-22
    assert d["registered"] and d["hasProcess"]


def test_bad_fft_size_throws_reference_message():
    _build()
    r = _node("""
      const m = require('./phaze_amd/node/phase-vocoder.js');
      const out = [];
      for (const n of [0, 1, 3, 1000]) {
        try { new m.PhaseVocoderProcessor({numberOfInputs: 1, numberOfOutputs: 1, processorOptions: {fftSize: n, hopSize: 1}}); out.push('no throw'); }
        catch (e) { out.push(e.message); }
      }
      console.log(JSON.stringify(out));""")
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == ["FFT size must be a power of two and bigger than 1"] * 4     # bundle:6-7


MAN = S.load_manifest()
CASES = {c["name"]: c for c in MAN["cases"]}
NODE_CASES = ["c2_1024_256_mono_pf1.5_tonal", "c3_2048_512_stereo_pf0.8_noise", "native_2048_128_mono_pf1.5_noise",
              "pause_1024_256_stereo_noise", "chanchange_1024_256_noise", "arate_1024_256_mono_tonal", "weird_pf_1024_256_mono_noise",
              "c5s_8192_2048_mono_sweep16_tonal", "outch_1024_256_mono_noise", "outch_2048_128_stereo_tonal",
              "stale_1024_256_stereo_tonal", "stale_2048_128_stereo_noise"]   # (the last two: an output channel that loses its input and gets it back, ola-processor.js:149-157)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 32], ids=["launch", "resident"])
@pytest.mark.parametrize("name", NODE_CASES)
def test_node_process_matches_reference_golden(name, flags, tmp_path):
    """PhaseVocoderProcessor.process() quantum by quantum against the reference's golden output -- launch per quantum, and with
    `processorOptions.flags = 32` (PV_FLAG_PERSISTENT_STREAM: the resident kernel where the shape has one, the launch form elsewhere)."""
    _build()
    case = CASES[name]
    h, T = case["hop"], case["store_hops"]
    nmax = S.case_max_channels(case)
    sig = np.stack([S.make_signal(case["signal"], ch, case["nhops"] * h)[:T * h] for ch in range(nmax)])
    pitch = S.pitch_schedule(case["pitch"], case["nhops"])[:T]
    sig.astype("<f4").tofile(tmp_path / "in.f32")
    pitch.astype("<f4").tofile(tmp_path / "pitch.f32")
    spec = {"fft": case["fft"], "hop": h, "nhops": T, "nch": case["nch"], "max_ch": nmax, "events": case.get("events", []),
            "arate": bool(case.get("arate")), "use_defaults": name.startswith("native") and not flags, "flags": flags,
            "in_file": str(tmp_path / "in.f32"), "pitch_file": str(tmp_path / "pitch.f32"), "out_file": str(tmp_path / "out.f32")}
    (tmp_path / "spec.json").write_text(json.dumps(spec))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node", "run_case.js"), str(tmp_path / "spec.json")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    got = np.fromfile(tmp_path / "out.f32", dtype="<f4").reshape(nmax, T * h)
    gold = S.load_golden_out(case)
    err = S.rms(got[:case["store_ch"]].astype(np.float64) - gold)
    assert np.all(np.isfinite(got))
    assert err < 2e-7, f"{name}: rms {err:.3e}"
    print(name, json.loads(r.stdout.strip().splitlines()[-1])["us_per_call"], "us/call")


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 32], ids=["launch", "resident"])
@pytest.mark.parametrize("case", MAN["multi_cases"], ids=lambda c: c["name"])
def test_node_two_inputs_match_reference_golden(case, flags, tmp_path):
    """numberOfInputs: 2 -- two inputs with different channel counts, one changing mid-stream (ola-processor.js:24-33,38-52): the host keeps
    one native handle per input; the reference's outputs for BOTH inputs are the golden."""
    _build()
    h, T = case["hop"], case["store_hops"]
    sig = S.multi_case_signals(case)
    pitch = S.pitch_schedule(case["pitch"], case["nhops"])[:T]
    pitch.astype("<f4").tofile(tmp_path / "pitch.f32")
    inputs = []
    for i, inp in enumerate(case["inputs"]):
        np.stack([s[:T * h] for s in sig[i]]).astype("<f4").tofile(tmp_path / f"in{i}.f32")
        inputs.append({"nch": inp["nch"], "events": inp.get("events", []), "max_ch": case["max_channels_per_input"][i], "in_file": str(tmp_path / f"in{i}.f32")})
    spec = {"fft": case["fft"], "hop": h, "nhops": T, "inputs": inputs, "pitch_file": str(tmp_path / "pitch.f32"), "out_file": str(tmp_path / "out.f32"), "flags": flags}
    (tmp_path / "spec.json").write_text(json.dumps(spec))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node", "run_multi.js"), str(tmp_path / "spec.json")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    got = np.fromfile(tmp_path / "out.f32", dtype="<f4")
    gold = np.concatenate([g.ravel() for g in S.load_golden_multi(case)])
    assert got.shape == gold.shape and np.all(np.isfinite(got))
    assert S.rms(got.astype(np.float64) - gold) < 2e-7


@pytest.mark.gpu
def test_wav_cli_shifts_pitch(tmp_path):
    """Node WAV-in/WAV-out CLI (counterpart of src/main.js): a 440 Hz tone at pitch 1.5 comes out near 660 Hz, same length and level;
    the streaming (process() per quantum) and batch entry points agree bit for bit."""
    import struct
    _build()
    rate, n = 48000, 48000
    t = np.arange(n) / rate
    x = (0.4 * np.sin(2 * np.pi * 440 * t) + S.make_signal("noise", 0, n) * 0.002).astype(np.float32)
    pcm = (np.clip(x, -1, 1) * 32767).astype("<i2").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, rate, rate * 2, 2, 16) + b"data" + struct.pack("<I", len(pcm))
    (tmp_path / "in.wav").write_bytes(hdr + pcm)
    cli = os.path.join(ROOT, "phaze_amd", "node", "pitch-shift-cli.js")
    outs = []
    for mode in ([], ["--batch"]):
        out = tmp_path / f"out{len(mode)}.wav"
        r = subprocess.run([NODE, cli, str(tmp_path / "in.wav"), str(out), "--pitch", "1.5", "--fft", "1024", "--hop", "256"] + mode,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr + r.stdout
        raw = out.read_bytes()
        y = np.frombuffer(raw[44:], dtype="<f4")
        assert y.shape[0] == n
        outs.append(y)
        spec = np.abs(np.fft.rfft(y[4096:4096 + 32768] * np.hanning(32768)))
        f_peak = np.argmax(spec) * rate / 32768
        assert 600 < f_peak < 720, f_peak
        assert 0.5 < S.rms(y[8192:-8192]) / S.rms(x[8192:-8192]) < 1.5
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.gpu
@pytest.mark.parametrize("fft,hop,streams,cps,shards,pf", [(4096, 1024, 16, 8, 2, 1.25), (1024, 256, 5, 2, 3, 0.8)])
def test_node_sharded_streams_in_flight_together(fft, hop, streams, cps, shards, pf, tmp_path):
    """Multi-GPU through the product boundary (independence: phase-vocoder.js:49-50,71): ONE Node process, stream s -> handle s mod G,
    every handle's batch started through the addon's asynchronous entry before the first wait.  On the one-GPU box the G handles share device 0
    (two kernels in flight on two HIP streams).  The gathered result equals ONE handle bit for bit and the oracle to 2e-7; the busy guard,
    the async/sync equivalence and a mid-stream migration through exportState / importState are checked on the way."""
    _build()
    T = 12
    x = np.stack([S.make_signal("tonal", c + 100 * s, T * hop) for s in range(streams) for c in range(cps)])
    p = np.stack([np.full(T, pf + 0.01 * s, np.float32) for s in range(streams)])
    x.astype("<f4").tofile(tmp_path / "in.f32")
    p.astype("<f4").tofile(tmp_path / "pitch.f32")
    spec = {"fft": fft, "hop": hop, "nhops": T, "streams": streams, "cps": cps, "shards": shards, "in_file": str(tmp_path / "in.f32"),
            "pitch_file": str(tmp_path / "pitch.f32"), "out_file": str(tmp_path / "out.f32")}
    (tmp_path / "spec.json").write_text(json.dumps(spec))
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "node", "run_sharded.js"), str(tmp_path / "spec.json")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["shards"] == shards and res["requested"] == shards and res["replicas"] == min(shards, res["devices"])
    assert res["equal_to_one_handle"] is True and res["async_equals_sync"] is True
    assert res["busy_guard"] == "PV_BUSY"
    assert res["state_cursor"] == T * hop and res["state_len"] == fft - hop and res["migrated_stream_continues_bit_exact"] is True
    assert res["in_place_equals_one_handle"] is True and res["thread_pool"] >= shards
    import oracle_lib
    got = np.fromfile(tmp_path / "out.f32", dtype="<f4").reshape(streams * cps, T * hop)
    for s in (0, streams - 1):
        ref = oracle_lib.Oracle(fft, hop, cps).process_planar(x[s * cps:(s + 1) * cps], p[s])
        assert S.rms(got[s * cps:(s + 1) * cps].astype(np.float64) - ref) < 2e-7


@pytest.mark.gpu
def test_node_sharded_bench_reports_what_it_measured():
    """tools/bench_node_sharded.js: the Node-side counterpart of `bench.py --gpus N` -- asks for 8 GPUs, measures the devices that exist and says so."""
    _build()
    r = subprocess.run([NODE, os.path.join(ROOT, "tools", "bench_node_sharded.js"), "--gpus", "8", "--streams", "16", "--channels", "2", "--fft", "1024", "--hop", "256",
                        "--hops", "64", "--steps", "3"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr + r.stdout
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["requested_gpus"] == 8 and j["replicas_measured"] == j["n_gpus"] == min(8, j["devices_present"]) and j["shards"] == 8
    assert j["frames_per_s"] > 1e4 and j["output_rms_stream0"] > 1e-3
    # 8 shards on the devices present: the libuv pool was raised to 8 before it started, so all eight batches were in flight together
    # (with the default pool of 4 at most four worker-thread windows overlap)
    assert j["config"]["uv_threadpool_size"] >= 8 and j["shards_in_flight_together"] >= 6, j


@pytest.mark.gpu
def test_node_two_inputs_cost_one_exposed_wait():
    """numberOfInputs: 2 launches both handles before it waits (processBegin / processEnd): the quantum must be clearly cheaper than the two
    launch + wait pairs of the sequential form (same process, same handles, measured back to back)."""
    _build()
    r = subprocess.run([NODE, os.path.join(ROOT, "tools", "bench_latency_node.js"), "600", "--inputs", "2"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr + r.stdout
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["inputs"] == 2 and j["sequential_p50"] is not None
    assert j["p50"] < 0.85 * j["sequential_p50"], (j["p50"], j["sequential_p50"])
