"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/phaze_amd.h declares, reports
reference-compatible errors without touching a device, and refuses to run without a GPU (no CPU compute path exists)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "phaze_amd.h")


def _declared():
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"PV_API\s+[\w\s\*]+?\b(pv_\w+)\s*\(", txt)))


def _lib():
    import phaze_amd
    if not os.path.exists(phaze_amd.library_path()):
        phaze_amd.build_library()
    return phaze_amd.load_library()


def test_header_declares_the_documented_surface():
    names = _declared()
    for must in ("pv_create", "pv_destroy", "pv_process", "pv_process_batch", "pv_process_batch_device", "pv_reset", "pv_reset_channels",
                 "pv_last_error", "pv_get_info", "pv_set_stream", "pv_synchronize", "pv_get_time_cursor", "pv_debug_frame", "pv_export_state",
                 "pv_import_state"):
        assert must in names
    assert "/root/reference/src/ola-processor.js:159-171" in open(HEADER).read()      # every entry point cites what it replaces


def test_library_exports_every_declared_symbol():
    L = _lib()
    missing = [n for n in _declared() if not hasattr(L, n)]
    assert not missing, missing
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "phaze_amd", "lib", "libphaze_amd.so")], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (pv_\w+)", out))
    assert set(_declared()) <= exported
    assert not [s for s in exported if "oracle" in s.lower()]                         # the checker is not linked into the product


def test_errors_without_device():
    import phaze_amd
    from phaze_amd import capi
    L = _lib()
    h = C.c_void_p()
    cfg = capi.make_config(1000, 250)
    assert L.pv_create(C.byref(cfg), C.byref(h)) == capi.PV_ERR_FFT_SIZE
    assert L.pv_last_error(None).decode() == "FFT size must be a power of two and bigger than 1"     # bundle:6-7
    cfg = capi.make_config(1024, 300)
    assert L.pv_create(C.byref(cfg), C.byref(h)) == capi.PV_ERR_ARGUMENT
    cfg = capi.make_config(2097152, 524288)
    assert L.pv_create(C.byref(cfg), C.byref(h)) == capi.PV_ERR_UNSUPPORTED
    assert L.pv_status_string(capi.PV_ERR_DEVICE).decode() == "HIP device error"
    # ABI guards (round 3): a config of another layout, or flag bits this build does not know, are refused before any device is touched
    assert L.pv_abi_version() == capi.ABI_VERSION == int(re.search(r"#define PV_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    cfg = capi.make_config(1024, 256)
    cfg.struct_size -= 4                                           # what a caller compiled against the round-2 header (no struct_size, 28 bytes) would pass
    assert L.pv_create(C.byref(cfg), C.byref(h)) == capi.PV_ERR_ARGUMENT and "struct_size" in L.pv_last_error(None).decode()
    cfg = capi.make_config(1024, 256, flags=0x400)
    assert L.pv_create(C.byref(cfg), C.byref(h)) == capi.PV_ERR_ARGUMENT and "flags" in L.pv_last_error(None).decode()
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not has_gpu:
        cfg = capi.make_config(1024, 256)
        assert L.pv_create(C.byref(cfg), C.byref(h)) == capi.PV_ERR_DEVICE                           # fails loudly: no CPU fallback
        with pytest.raises(phaze_amd.PvError):
            phaze_amd.PhaseVocoder(fft_size=1024, hop_size=256)


def test_product_sources_do_not_reference_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "phaze_amd")):
        for f in files:
            if f.endswith((".hip", ".h", ".c", ".py", ".js")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "pv_oracle" not in txt and "oracle_lib" not in txt and "libpv_oracle" not in txt, os.path.join(base, f)


def test_no_environment_switches_in_the_product_library():
    """Round-1 builds read PHAZE_ABLATE / PHAZE_GENERIC_KERNEL / PHAZE_STREAM_COPY from the environment; a stray variable could select a
    work-skipping kernel instance.  The shipped library contains no ablation code and never calls getenv: A/B switches are explicit
    pv_config.flags bits."""
    lib = os.path.join(ROOT, "phaze_amd", "lib", "libphaze_amd.so")
    _lib()
    strings = subprocess.run(["strings", "-a", lib], capture_output=True, text=True).stdout
    assert "PHAZE_" not in strings
    undefined = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True).stdout
    assert "getenv" not in undefined
    for base, _, files in os.walk(os.path.join(ROOT, "phaze_amd", "csrc")):
        for f in files:
            txt = open(os.path.join(base, f), errors="ignore").read()
            assert "getenv" not in txt and "ablate" not in txt, f


def test_flag_constants_match_the_header():
    """The ctypes binding's FLAG_* values are the header's PV_FLAG_* enumerators."""
    hdr = open(os.path.join(ROOT, "include", "phaze_amd.h")).read()
    vals = dict(re.findall(r"(PV_FLAG_[A-Z0-9_]+)\s*=\s*(\d+)", hdr))
    assert vals == {"PV_FLAG_GENERIC_KERNEL": "1", "PV_FLAG_STREAM_COPY": "2", "PV_FLAG_WORKGROUP_KERNEL": "4", "PV_FLAG_STREAM_EVENT_WAIT": "8", "PV_FLAG_STREAM_PINNED_INPUT": "16",
                    "PV_FLAG_PERSISTENT_STREAM": "32", "PV_FLAG_TEST_NO_HDP_FLUSH": "64", "PV_FLAG_HOST_CHANNEL_BOOKKEEPING": "128", "PV_FLAG_FP64_FORWARD": "256", "PV_FLAG_TEST_FAIL_SECOND_PIECE": "512", "PV_FLAG_ALL": "1023"}
    import phaze_amd
    assert (phaze_amd.FLAG_GENERIC_KERNEL, phaze_amd.FLAG_STREAM_COPY, phaze_amd.FLAG_WORKGROUP_KERNEL, phaze_amd.FLAG_STREAM_EVENT_WAIT,
            phaze_amd.FLAG_STREAM_PINNED_INPUT) == (1, 2, 4, 8, 16)
    assert phaze_amd.FLAG_FP64_FORWARD == 256
