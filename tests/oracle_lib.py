"""ctypes binding of the CPU oracle (oracle/libpv_oracle.so).  TEST INFRASTRUCTURE ONLY.

The oracle is the checker the HIP path is compared against; nothing under phaze_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "libpv_oracle.so")
_lib = None


def build_oracle(force: bool = False) -> str:
    src = os.path.join(_ORACLE_DIR, "pv_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-B", "libpv_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_oracle())
        fp = C.POINTER(C.c_float)
        L.pvo_create.restype = C.c_void_p
        L.pvo_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.pvo_destroy.argtypes = [C.c_void_p]
        L.pvo_process.restype = C.c_int
        L.pvo_process.argtypes = [C.c_void_p, C.POINTER(fp), C.POINTER(fp), C.c_int, C.c_int, C.c_float]
        L.pvo_process2.restype = C.c_int
        L.pvo_process2.argtypes = [C.c_void_p, C.POINTER(fp), C.POINTER(fp), C.c_int, C.c_int, C.c_int, C.c_float]
        L.pvo_process_planar.restype = C.c_int
        L.pvo_process_planar.argtypes = [C.c_void_p, fp, fp, C.c_int, C.c_int, C.c_long, fp]
        for name, rt in (("pvo_debug_X", C.POINTER(C.c_double)), ("pvo_debug_Y", C.POINTER(C.c_double)),
                         ("pvo_debug_mag", fp), ("pvo_debug_peaks", C.POINTER(C.c_int32))):
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [C.c_void_p]
        L.pvo_debug_npeaks.restype = C.c_int
        L.pvo_debug_npeaks.argtypes = [C.c_void_p]
        L.pvo_time_cursor.restype = C.c_double
        L.pvo_time_cursor.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Oracle:
    """One reference processor instance (one input / one output, `nch` channels)."""

    def __init__(self, fft_size: int, hop: int, nch: int = 1):
        self.h = lib().pvo_create(fft_size, hop, nch)
        if not self.h:
            raise ValueError("FFT size must be a power of two and bigger than 1")
        self.N, self.hop = fft_size, hop

    def close(self):
        if self.h:
            lib().pvo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def process(self, blocks, pitch: float, paused: bool = False, nout: int = -1):
        """blocks: list of float32[hop] per channel -> list of float32[hop] outputs.  nout: outputs[0].length when it differs from the input channel
        count (ola-processor.js:46-51 reallocates the output buffers on its own); only the input-count channels are written (ola:111-118)."""
        nch = len(blocks)
        ins = [np.ascontiguousarray(b, dtype=np.float32) for b in blocks]
        outs = [np.zeros(self.hop, dtype=np.float32) for _ in range(nch)]
        fp = C.POINTER(C.c_float)
        ip = (fp * max(nch, 1))(*[_fptr(a) for a in ins])
        op = (fp * max(nch, 1))(*[_fptr(a) for a in outs])
        ok = lib().pvo_process2(self.h, ip, op, nch, nch if nout < 0 else nout, 1 if paused else 0, C.c_float(float(pitch)))
        if not ok:
            raise TypeError("outputs[0] has fewer channels than inputs[0] (the reference throws here)")
        return outs

    def process_planar(self, x: np.ndarray, pitch: np.ndarray) -> np.ndarray:
        """x: [nch, nhops*hop] float32, pitch: [nhops] float32 -> same shape output."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        nch, n = x.shape
        nhops = n // self.hop
        pitch = np.ascontiguousarray(pitch, dtype=np.float32)
        assert pitch.shape[0] >= nhops
        y = np.zeros_like(x)
        lib().pvo_process_planar(self.h, _fptr(x), _fptr(y), nch, nhops, n, _fptr(pitch))
        return y

    def debug(self) -> dict:
        L, N = lib(), self.N
        H = N // 2 + 1
        npk = L.pvo_debug_npeaks(self.h)
        return {
            "X": np.ctypeslib.as_array(L.pvo_debug_X(self.h), shape=(2 * N,)).copy(),
            "Y": np.ctypeslib.as_array(L.pvo_debug_Y(self.h), shape=(2 * N,)).copy(),
            "mag": np.ctypeslib.as_array(L.pvo_debug_mag(self.h), shape=(H,)).copy(),
            "peaks": np.ctypeslib.as_array(L.pvo_debug_peaks(self.h), shape=(H,))[:npk].copy(),
        }


def run_case(case: dict, signals, pitch, collect_dumps=False):
    """Drive one golden case (incl. pause / channel-change events) through the oracle.
    Returns (out[maxch, nhops*hop], dumps{hop: debug dict})."""
    N, h, T = case["fft"], case["hop"], case["nhops"]
    nch = case["nch"]
    o = Oracle(N, h, nch)
    out = np.zeros((len(signals), T * h), dtype=np.float32)
    dumps = {}
    nout = -1
    for m in range(T):
        paused = False
        for e in case.get("events", []):
            if e["hop"] == m:
                if e["type"] == "pause":
                    paused = True
                if e["type"] == "channels":
                    nch = e["nch"]
                if e["type"] == "out_channels":
                    nout = e["nch"]
        blocks = [signals[c][m * h:(m + 1) * h] for c in range(nch)]
        res = o.process(blocks, pitch[m], paused, nout=max(nout, nch) if nout >= 0 else -1)
        for c in range(nch):
            out[c, m * h:(m + 1) * h] = res[c]
        if collect_dumps and m in case.get("dump_hops", []):
            dumps[m] = o.debug()
    o.close()
    return out, dumps


def run_multi_case(case: dict, signals, pitch):
    """numberOfInputs > 1 (ola-processor.js:10-11,24-33): one processor state per input, each with its own channel list; a channel-count
    change reallocates only that input (ola:38-52).  Returns a list (per input) of out[maxch_i, nhops*hop]."""
    N, h, T = case["fft"], case["hop"], case["nhops"]
    nin = len(case["inputs"])
    nch = [inp["nch"] for inp in case["inputs"]]
    procs = [Oracle(N, h, n) for n in nch]
    outs = [np.zeros((len(signals[i]), T * h), dtype=np.float32) for i in range(nin)]
    for m in range(T):
        for i in range(nin):
            for e in case["inputs"][i].get("events", []):
                if e["hop"] == m and e["type"] == "channels":
                    nch[i] = e["nch"]
            res = procs[i].process([signals[i][c][m * h:(m + 1) * h] for c in range(nch[i])], pitch[m])
            for c in range(nch[i]):
                outs[i][c, m * h:(m + 1) * h] = res[c]
    for p_ in procs:
        p_.close()
    return outs
