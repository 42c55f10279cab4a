"""Seeded synthetic signals of SURVEY.md section 4, bit-identical to tests/golden/gen_golden.js.

Test infrastructure: inputs are never stored, they are regenerated from seeds here (numpy) and in the JS
golden generator; manifest.json carries the sha256 of channel 0 so a drift between the two is caught.
"""
import hashlib
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_A, _C = 1664525, 1013904223


def lcg_u32(seed: int, n: int) -> np.ndarray:
    """s_{i+1} = (s_i*1664525 + 1013904223) mod 2^32, returns s_1..s_n (block-doubling jump-ahead)."""
    out = np.empty(max(n, 1), dtype=np.uint64)
    m = np.uint64(0xFFFFFFFF)
    out[0] = (np.uint64(seed & 0xFFFFFFFF) * np.uint64(_A) + np.uint64(_C)) & m
    a, c, k = np.uint64(_A), np.uint64(_C), 1
    with np.errstate(over="ignore"):
        while k < n:
            cnt = min(k, n - k)
            out[k:k + cnt] = (out[:cnt] * a + c) & m
            c = (a * c + c) & m
            a = (a * a) & m
            k *= 2
    return out[:n].astype(np.uint32)


def lcg_noise(seed: int, n: int, amp: float) -> np.ndarray:
    s = lcg_u32(seed, n)
    return (((s >> np.uint32(8)).astype(np.float64) - 8388608.0) / 8388608.0 * amp).astype(np.float32)


def _tri(i, P):
    return 4.0 * np.abs((i % P) / float(P) - 0.5) - 1.0


def make_signal(kind: str, ch: int, n: int, stream: int = 0) -> np.ndarray:
    off = 100000 * stream
    if kind == "noise":
        return lcg_noise(1000 + ch + off, n, 0.5)
    if kind == "tonal":
        nz = lcg_noise(2000 + ch + off, n, 1.0 / 64).astype(np.float64)
        i = np.arange(n, dtype=np.float64)
        x = 0.25 * _tri(i, 109) + 0.125 * _tri(i, 31) + 0.0625 * _tri(i, 7) + nz
        return x.astype(np.float32)
    if kind == "impulse":
        x = np.zeros(n, dtype=np.float32)
        x[1000 + ch] = 1.0
        return x
    if kind == "sine32":
        i = np.arange(n)
        return (0.5 * np.sin(2 * np.pi * (i % 32) / 32)).astype(np.float32)
    if kind == "silence":
        return np.zeros(n, dtype=np.float32)
    raise ValueError(kind)


def pitch_schedule(spec: dict, nhops: int) -> np.ndarray:
    if "const" in spec:
        return np.full(nhops, spec["const"], dtype=np.float32)
    if "sweep" in spec:
        a, b, n = spec["sweep"]
        i = np.arange(nhops, dtype=np.float64)
        return (a + (b - a) * i / (n - 1)).astype(np.float32)
    if "list" in spec:
        with np.errstate(over="ignore"):
            return np.array([float(v) for v in spec["list"][:nhops]], dtype=np.float64).astype(np.float32)
    raise ValueError(spec)


def sha256_hex(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_manifest() -> dict:
    """Single-input cases under "cases" (what most tests iterate over); numberOfInputs > 1 cases under "multi_cases"."""
    with open(os.path.join(GOLDEN_DIR, "manifest.json")) as f:
        m = json.load(f)
    m["multi_cases"] = [c for c in m["cases"] if "inputs" in c]
    m["cases"] = [c for c in m["cases"] if "inputs" not in c]
    return m


def multi_case_signals(case: dict):
    """signals[i][c]: channel c of input i uses seed channel 10*i + c (tests/golden/gen_golden.js runMultiCase)."""
    n = case["nhops"] * case["hop"]
    return [[make_signal(case["signal"], 10 * i + c, n) for c in range(mc)] for i, mc in enumerate(case["max_channels_per_input"])]


def load_golden_multi(case: dict):
    """Returns a list (per input) of [max channels of that input, store_hops*hop] float32."""
    a = np.fromfile(os.path.join(GOLDEN_DIR, case["out_file"]), dtype="<f4")
    n = case["store_hops"] * case["hop"]
    out, pos = [], 0
    for mc in case["max_channels_per_input"]:
        out.append(a[pos:pos + mc * n].reshape(mc, n))
        pos += mc * n
    return out


def load_golden_out(case: dict) -> np.ndarray:
    """Returns [store_ch, store_hops*hop] float32."""
    a = np.fromfile(os.path.join(GOLDEN_DIR, case["out_file"]), dtype="<f4")
    return a.reshape(case["store_ch"], case["store_hops"] * case["hop"])


def load_dump(case: dict, dump: dict) -> dict:
    raw = open(os.path.join(GOLDEN_DIR, dump["file"]), "rb").read()
    out, pos = {}, 0
    dt = {"f64": "<f8", "f32": "<f4", "i32": "<i4"}
    for name, ty, cnt in dump["layout"]:
        d = np.dtype(dt[ty])
        out[name] = np.frombuffer(raw, dtype=d, count=cnt, offset=pos).copy()
        pos += d.itemsize * cnt
    return out


def case_max_channels(case: dict) -> int:
    return max([case["nch"]] + [e["nch"] for e in case.get("events", []) if e["type"] == "channels"])


def rms(x) -> float:
    x = np.asarray(x, dtype=np.float64)
    return float(np.sqrt(np.mean(x * x))) if x.size else 0.0
