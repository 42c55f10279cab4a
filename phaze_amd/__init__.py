"""phaze_amd -- MI355X-native phase-vocoder pitch shifter (the process() path of olvb/phaze).

The product is the gfx950 shared library `phaze_amd/lib/libphaze_amd.so` behind the C ABI of
`include/phaze_amd.h`, plus the Node.js host in `phaze_amd/node/`.  This Python package is only the thin
ctypes binding tests and bench.py use; it contains no compute path and raises if the library is missing.
"""
from .capi import (FLAG_FP64_FORWARD, FLAG_GENERIC_KERNEL, FLAG_HOST_CHANNEL_BOOKKEEPING, FLAG_PERSISTENT_STREAM, FLAG_STREAM_COPY, FLAG_STREAM_EVENT_WAIT, FLAG_STREAM_PINNED_INPUT, FLAG_TEST_FAIL_SECOND_PIECE, FLAG_TEST_NO_HDP_FLUSH,  # noqa: F401
                   FLAG_WORKGROUP_KERNEL, PhaseVocoder, PvError, build_library, library_path, load_library, pinned_empty)

__all__ = ["PhaseVocoder", "PvError", "build_library", "library_path", "load_library", "FLAG_GENERIC_KERNEL", "FLAG_STREAM_COPY", "FLAG_WORKGROUP_KERNEL",
           "FLAG_STREAM_PINNED_INPUT", "FLAG_STREAM_EVENT_WAIT", "FLAG_PERSISTENT_STREAM", "FLAG_TEST_NO_HDP_FLUSH", "FLAG_TEST_FAIL_SECOND_PIECE", "FLAG_HOST_CHANNEL_BOOKKEEPING", "FLAG_FP64_FORWARD", "pinned_empty"]
