"""ctypes binding of include/phaze_amd.h (the same entry points the N-API addon binds).

Mirrors the reference surface: `PhaseVocoder.process(inputs, outputs, parameters)` has the argument
meaning of OLAProcessor.process (/root/reference/src/ola-processor.js:159-171).  There is NO fallback:
if libphaze_amd.so cannot be loaded, or no HIP device exists, construction raises.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("PHAZE_LIB") or os.path.join(_HERE, "lib", "libphaze_amd.so")   # PHAZE_LIB: A/B builds of the same ABI
_lib = None

PV_OK, PV_ERR_FFT_SIZE, PV_ERR_ARGUMENT, PV_ERR_UNSUPPORTED, PV_ERR_CAPACITY, PV_ERR_DEVICE, PV_ERR_DESTROYED = range(7)

EXPORTS = [
    "pv_create", "pv_destroy", "pv_last_error", "pv_status_string", "pv_get_info", "pv_reset", "pv_reset_channels",
    "pv_get_time_cursor", "pv_set_time_cursor", "pv_process", "pv_process_batch", "pv_process_batch_device",
    "pv_set_stream", "pv_synchronize", "pv_debug_frame", "pv_export_state", "pv_import_state", "pv_abi_version",
    "pv_process_begin", "pv_process_end", "pv_device_count", "pv_host_alloc", "pv_host_free", "pv_reset_channels_part",
    "pv_forward_stats",
]


class PvError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


class _Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("struct_size", "fft_size", "hop_size", "max_channels", "max_hops", "device_id", "frames_per_chunk", "flags")]


def make_config(fft_size, hop_size, max_channels=1, max_hops=1, device_id=0, frames_per_chunk=0, flags=0):
    """pv_config with struct_size filled in (the C side's PV_CONFIG_INIT)."""
    return _Config(C.sizeof(_Config), fft_size, hop_size, max_channels, max_hops, device_id, frames_per_chunk, flags)


ABI_VERSION = 5          # PV_ABI_VERSION of include/phaze_amd.h this binding was written against (checked at load time)


# pv_config.flags (include/phaze_amd.h): explicit A/B switches; the library reads no environment variables
FLAG_GENERIC_KERNEL, FLAG_STREAM_COPY, FLAG_WORKGROUP_KERNEL, FLAG_STREAM_EVENT_WAIT, FLAG_STREAM_PINNED_INPUT, FLAG_PERSISTENT_STREAM = 1, 2, 4, 8, 16, 32
FLAG_TEST_NO_HDP_FLUSH = 64      # test hook (tests/test_gpu_stream_forms.py)
FLAG_HOST_CHANNEL_BOOKKEEPING = 128
FLAG_TEST_FAIL_SECOND_PIECE = 512   # test hook (tests/test_gpu_batch_pipeline.py): a pipelined host-buffer batch fails behind its second piece
FLAG_FP64_FORWARD = 256          # every forward transform in fp64 (the round-4 kernels); default: fp32 first, fp64 only where a peak decision is in doubt
STATE_HISTORY, STATE_ACCUMULATOR = 1, 2


class _Info(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("fft_size", "hop_size", "overlaps", "max_channels", "max_hops", "threads_per_workgroup",
                                          "lds_bytes_per_workgroup", "frames_per_chunk", "compute_units", "device_id")] + [("device_name", C.c_char * 64),
                                                                                                       ("kernel_name", C.c_char * 32)]


def library_path() -> str:
    return _LIB_PATH


def build_library(force: bool = False) -> str:
    """Compile the gfx950 library in-tree with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    args = ["make", "-C", csrc]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64; two HIP runtimes in one process cannot both open the GPU ("no ROCm-capable
    # device").  If torch is installed, load it FIRST so that this library binds to the runtime torch already mapped (same SONAME).
    if "torch" not in sys.modules and not os.environ.get("PHAZE_NO_TORCH_PRELOAD"):
        try:
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(_LIB_PATH):
        raise PvError(PV_ERR_DEVICE, f"{_LIB_PATH} is missing: build it with phaze_amd.build_library() / __graft_entry__.build(); "
                                     "there is no CPU fallback")
    L = C.CDLL(_LIB_PATH)
    fp, vp = C.POINTER(C.c_float), C.c_void_p
    L.pv_create.argtypes = [C.POINTER(_Config), C.POINTER(vp)]
    L.pv_destroy.argtypes = [vp]
    L.pv_last_error.argtypes = [vp]
    L.pv_last_error.restype = C.c_char_p
    L.pv_status_string.argtypes = [C.c_int]
    L.pv_status_string.restype = C.c_char_p
    L.pv_get_info.argtypes = [vp, C.POINTER(_Info)]
    L.pv_reset.argtypes = [vp]
    L.pv_reset_channels.argtypes = [vp, C.c_int32, C.c_int32]
    L.pv_reset_channels_part.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    L.pv_get_time_cursor.argtypes = [vp, C.POINTER(C.c_int64)]
    L.pv_set_time_cursor.argtypes = [vp, C.c_int64]
    L.pv_process.argtypes = [vp, C.POINTER(fp), C.POINTER(fp), C.c_int32, C.c_int32, C.c_float]
    L.pv_process_batch.argtypes = [vp, fp, fp, C.c_int32, C.c_int32, C.c_int64, fp, C.c_int32, C.c_int32]
    L.pv_process_batch_device.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, C.c_int64, vp, C.c_int32, C.c_int32]
    L.pv_set_stream.argtypes = [vp, vp]
    L.pv_synchronize.argtypes = [vp]
    L.pv_debug_frame.argtypes = [vp, C.c_int32, fp, C.c_float, C.POINTER(C.c_double), fp, C.POINTER(C.c_int32), fp]
    L.pv_export_state.argtypes = [vp, C.c_int32, fp, fp, C.POINTER(C.c_int64)]
    L.pv_import_state.argtypes = [vp, C.c_int32, fp, fp, C.c_int64]
    for n in EXPORTS:
        if n not in ("pv_last_error", "pv_status_string"):
            getattr(L, n).restype = C.c_int
    L.pv_process_begin.argtypes = [vp, C.POINTER(fp), C.c_int32, C.c_int32, C.c_float]
    L.pv_process_end.argtypes = [vp, C.POINTER(fp)]
    L.pv_device_count.argtypes = [C.POINTER(C.c_int32)]
    L.pv_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.pv_host_free.argtypes = [vp]
    L.pv_forward_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32]
    L.pv_abi_version.argtypes = []
    if L.pv_abi_version() != ABI_VERSION:
        raise PvError(PV_ERR_ARGUMENT, f"{_LIB_PATH} has PV_ABI_VERSION {L.pv_abi_version()}, this binding expects {ABI_VERSION}: rebuild the library")
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class _PinnedBlock:
    """Owner of one pv_host_alloc block: frees it when the last numpy view is gone."""

    def __init__(self, nbytes):
        self._L = load_library()
        self.ptr = C.c_void_p()
        rc = self._L.pv_host_alloc(max(int(nbytes), 1), C.byref(self.ptr))
        if rc != PV_OK:
            raise PvError(rc, self._L.pv_last_error(None).decode())
        self.nbytes = int(nbytes)

    def __del__(self):
        try:
            if self.ptr and self.ptr.value:
                self._L.pv_host_free(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32):
    """numpy array in page-locked host memory (pv_host_alloc): process_batch() pipelines batches that live in such arrays (DMA both ways at
    once).  The memory is released when the array and every view of it are gone."""
    shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if shape else 1
    blk = _PinnedBlock(n * dt.itemsize)
    buf = (C.c_char * max(n * dt.itemsize, 1)).from_address(blk.ptr.value)
    buf._pv_owner = blk                              # the ctypes object keeps the block alive; numpy keeps the ctypes object alive (base)
    return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)


class PhaseVocoder:
    """One processor instance = one `new PhaseVocoderProcessor(options)` (phase-vocoder.js:24-43)."""

    parameter_descriptors = [{"name": "pitchFactor", "defaultValue": 1.0}]   # phase-vocoder.js:17-22

    def __init__(self, fft_size=2048, hop_size=128, max_channels=2, max_hops=1, device_id=0, frames_per_chunk=0, flags=0):
        self._L = load_library()
        self._h = C.c_void_p()
        cfg = make_config(fft_size, hop_size, max_channels, max_hops, device_id, frames_per_chunk, flags)
        rc = self._L.pv_create(C.byref(cfg), C.byref(self._h))
        if rc != PV_OK:
            msg = self._L.pv_last_error(None).decode()
            self._h = C.c_void_p()
            if rc == PV_ERR_FFT_SIZE:
                raise ValueError(msg)           # the reference throws Error('FFT size must be ...') (bundle:6-7)
            raise PvError(rc, msg)
        self.fft_size, self.hop_size = fft_size, hop_size
        self.max_channels, self.max_hops = max_channels, max_hops
        self._host_channels = bool(flags & FLAG_HOST_CHANNEL_BOOKKEEPING)
        self._nin = self._nout = 1                          # "default to 1 channel per input / output until we know more" (ola-processor.js:24-33)
        # host bookkeeping only: what it takes to reproduce an output channel that lost its input while outputs[0].length stayed (see _stale_frame)
        self._device_id, self._flags = device_id, flags
        self._window = np.zeros((max_channels, fft_size), np.float32) if self._host_channels else None   # inputBuffers as the reference holds them (ola-processor.js:59,121-127)
        self._stale = {}                                    # channel -> [frame / nbOverlaps as the last real quantum added it, quanta it has been re-added since]
        self._last_pitch = None
        self._scratch = None

    # -- lifetime --
    def close(self):
        if getattr(self, "_scratch", None) is not None:
            self._scratch.close()
            self._scratch = None
        if getattr(self, "_h", None) and self._h.value:
            self._L.pv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != PV_OK:
            raise PvError(rc, self._L.pv_last_error(self._h).decode())

    # -- state --
    def reset(self):
        self._check(self._L.pv_reset(self._h))

    def reset_channels(self, first, count, parts=STATE_HISTORY | STATE_ACCUMULATOR):
        self._check(self._L.pv_reset_channels_part(self._h, first, count, parts))

    @property
    def time_cursor(self):
        v = C.c_int64()
        self._check(self._L.pv_get_time_cursor(self._h, C.byref(v)))
        return v.value

    @time_cursor.setter
    def time_cursor(self, value):
        self._check(self._L.pv_set_time_cursor(self._h, int(value)))

    def export_state(self, ch):
        """(hist[N-hop], acc[N-hop], time_cursor) of channel slot `ch`: ola-processor.js:59,77 + phase-vocoder.js:31."""
        L = self.fft_size - self.hop_size
        hist, acc, tc = np.zeros(max(L, 1), np.float32), np.zeros(max(L, 1), np.float32), C.c_int64()
        self._check(self._L.pv_export_state(self._h, ch, _fp(hist), _fp(acc), C.byref(tc)))
        return hist[:L], acc[:L], tc.value

    def import_state(self, ch, hist=None, acc=None, time_cursor=-1):
        L = self.fft_size - self.hop_size
        def chk(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.size != L:
                raise ValueError(f"state arrays hold N - hop = {L} floats")
            return a
        hist, acc = chk(hist), chk(acc)
        self._check(self._L.pv_import_state(self._h, ch, _fp(hist) if hist is not None and L else None,
                                            _fp(acc) if acc is not None and L else None, C.c_int64(int(time_cursor))))

    def info(self):
        i = _Info()
        self._check(self._L.pv_get_info(self._h, C.byref(i)))
        d = {n: getattr(i, n) for n, _ in _Info._fields_}
        d["device_name"] = i.device_name.decode()
        d["kernel_name"] = i.kernel_name.decode()
        return d

    # -- the hot call, AudioWorklet form --
    def process(self, inputs, outputs, parameters):
        """inputs[0][c]: float32[hop] (or length 0 when paused); outputs[0][c]: float32[hop], filled.
        parameters['pitchFactor']: float32 array, last element used (phase-vocoder.js:47).  Returns True."""
        chans = inputs[0]
        nch = len(chans)
        if self._host_channels:
            # the reference's reallocateChannelsIfNeeded (ola-processor.js:38-52): inputs and outputs are two separate events
            if len(outputs[0]) < nch:
                raise TypeError("outputs[0] has fewer channels than inputs[0]: the reference's processOLA dereferences outputs[i][j] (phase-vocoder.js:51) and throws")
            out_changed = len(outputs[0]) != self._nout
            if nch != self._nin:
                if not out_changed:
                    # One corner of ola-processor.js (149-157): `outputBuffersToRetrieve` is only reallocated with the OUTPUT channels, so an output channel whose input
                    # disappears keeps its last frame there, and handleOutputBuffersToRetrieve goes on adding that stale frame (and shifting) every quantum.  Nobody
                    # hears it (writeOutputs walks the input channels) -- unless the input returns before the output count changes: then the channel's pending sums
                    # are those of the stale frame.  Lost channels: remember the frame; regained channels: their accumulator becomes what the reference's has become.
                    for c in range(nch, min(self._nin, self._nout, self.max_channels)):
                        self._stale[c] = [self._stale_frame(c), 0]
                    regained = {c: self._stale_accumulator(c) for c in range(self._nin, min(nch, self.max_channels)) if c in self._stale}
                else:
                    regained = {}
                self.reset_channels(0, self.max_channels, STATE_HISTORY)
                self._window[:] = 0.0
                for c, acc in regained.items():
                    self.import_state(c, acc=acc)
                    del self._stale[c]
                self._nin = nch
            if out_changed:
                self._stale.clear()                                              # allocateOutputChannels: fresh (zeroed) outputBuffersToRetrieve (ola-processor.js:74-85)
                self.reset_channels(0, self.max_channels, STATE_ACCUMULATOR)
                self._nout = len(outputs[0])
        pf = np.asarray(parameters["pitchFactor"], dtype=np.float32)
        pitch = float(pf[-1])
        paused = nch > 0 and len(chans[0]) == 0                                 # ola-processor.js:93
        outs = outputs[0]
        fpt = C.POINTER(C.c_float)
        keep = [np.ascontiguousarray(c, dtype=np.float32) for c in chans]
        ip = (fpt * max(nch, 1))(*[_fp(a) if a.size else None for a in keep])
        tmp = [np.zeros(self.hop_size, dtype=np.float32) for _ in range(nch)]
        op = (fpt * max(nch, 1))(*[_fp(a) for a in tmp])
        self._check(self._L.pv_process(self._h, ip, op, nch, 0 if paused else self.hop_size, C.c_float(pitch)))
        for c in range(min(nch, len(outs))):
            outs[c][:] = tmp[c]
        if self._host_channels:
            h = self.hop_size
            for c in range(min(nch, self.max_channels)):                           # readInputs + shiftInputBuffers (ola-processor.js:89-127)
                w = self._window[c]
                if h < self.fft_size:
                    w[:-h] = w[h:].copy()
                w[-h:] = 0.0 if paused else keep[c]
            self._last_pitch = pitch
            for rec in self._stale.values():
                rec[1] += 1
        return True

    def _stale_frame(self, ch):
        """The windowed frame / nbOverlaps that the last quantum added for channel `ch`, recomputed on a one-channel scratch handle from the window the reference's
        inputBuffers held (zero accumulator in: the hop that comes out and the accumulator left behind ARE the frame, 0 + x being exact)."""
        N, h = self.fft_size, self.hop_size
        t = self.time_cursor
        if self._last_pitch is None or t < h:
            return np.zeros(N, np.float32)                                        # no frame yet: outputBuffersToRetrieve still holds its zeros
        if self._scratch is None:
            self._scratch = PhaseVocoder(N, h, 1, 1, self._device_id, 0, self._flags & FLAG_FP64_FORWARD)
        sc = self._scratch
        sc.import_state(0, hist=self._window[ch, :N - h], acc=np.zeros(N - h, np.float32), time_cursor=t - h)
        out = [[np.zeros(h, np.float32)]]
        sc.process([[self._window[ch, N - h:].copy()]], out, {"pitchFactor": np.array([self._last_pitch], np.float32)})
        return np.concatenate([out[0][0], sc.export_state(0)[1]]).astype(np.float32)

    def _stale_accumulator(self, ch):
        """outputBuffers of a channel whose stale frame has been re-added for q quanta (f32 adds in the reference's order, ola-processor.js:130-157)."""
        N, h = self.fft_size, self.hop_size
        frame, q = self._stale[ch]
        a = np.concatenate([self.export_state(ch)[1], np.zeros(h, np.float32)]).astype(np.float32)
        for _ in range(min(q, N // h + 1)):                                       # (after N / hop quanta the sums no longer change)
            a = a + frame
            a = np.concatenate([a[h:], np.zeros(h, np.float32)])
        return a[:N - h]

    # -- the hot call, batch forms --
    def process_batch(self, x, pitch, channels_per_stream=0, out=None):
        """x: float32[nch, nhops*hop] (host); pitch: float32[nhops] or [nstreams, nhops]. Returns y like x (written into `out` when given:
        with x and out in page-locked memory -- pinned_empty() -- the call is pipelined, see pv_process_batch)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        nch, n = x.shape
        nhops = n // self.hop_size
        assert nhops * self.hop_size == n
        pitch = np.ascontiguousarray(pitch, dtype=np.float32)
        stride = 0
        if pitch.ndim == 2:
            stride = pitch.shape[1]
        if out is not None:
            assert out.shape == x.shape and out.dtype == np.float32 and out.flags.c_contiguous
            y = out
        else:
            y = np.empty_like(x)
        self._check(self._L.pv_process_batch(self._h, _fp(x), _fp(y), nch, nhops, n, _fp(pitch), stride, channels_per_stream or 1))
        return y

    def process_batch_device(self, d_in, d_out, nch, nhops, ch_stride, d_pitch, pitch_stride=0, channels_per_stream=1):
        """Raw device pointers (ints).  Asynchronous on the handle's stream."""
        self._check(self._L.pv_process_batch_device(self._h, C.c_void_p(d_in), C.c_void_p(d_out), nch, nhops, ch_stride,
                                                    C.c_void_p(d_pitch), pitch_stride, channels_per_stream))

    def set_stream(self, hip_stream):
        self._check(self._L.pv_set_stream(self._h, C.c_void_p(hip_stream)))

    def synchronize(self):
        self._check(self._L.pv_synchronize(self._h))

    def forward_stats(self, reset=False):
        """(frames whose forward transform an fp32-first instance computed, frames of those that re-ran it in fp64) since creation / the last reset."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self._L.pv_forward_stats(self._h, C.byref(a), C.byref(b), 1 if reset else 0))
        return int(a.value), int(b.value)

    # -- test tap --
    def debug_frame(self, ch, block, pitch):
        N = self.fft_size
        H = N // 2 + 1
        block = np.ascontiguousarray(block, dtype=np.float32)
        X = np.zeros(2 * N, dtype=np.float64)
        mag = np.zeros(H, dtype=np.float32)
        flags = np.zeros(H, dtype=np.int32)
        Y = np.zeros(2 * H, dtype=np.float32)
        self._check(self._L.pv_debug_frame(self._h, ch, _fp(block), C.c_float(float(pitch)), X.ctypes.data_as(C.POINTER(C.c_double)),
                                           _fp(mag), flags.ctypes.data_as(C.POINTER(C.c_int32)), _fp(Y)))
        return {"X": X, "mag": mag, "flags": flags, "Y": Y}
