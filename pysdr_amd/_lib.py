"""ctypes binding of libpysdr_hip.so (include/pysdr_hip.h).

There is NO CPU fallback: if the library is missing, or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# the diagnostic build (work-skipping ablation switches compiled in) is a different file and is
# only ever loaded on explicit request
# ... and so is an A/B variant (python -m pysdr_amd.build --variant NAME): only under the tuning master switch
_variant = os.environ.get("PYSDR_LIB_VARIANT", "") if os.environ.get("PYSDR_TUNING", "0") not in ("", "0") else ""
LIB_PATH = os.path.join(HERE, "libpysdr_hip_diag.so" if os.environ.get("PYSDR_USE_DIAG_LIB") == "1"
                        else (f"libpysdr_hip_{_variant}.so" if _variant else "libpysdr_hip.so"))

MAX_RX = 8


class PysdrError(RuntimeError):
    pass


class Cfg(C.Structure):
    _fields_ = [("srate", C.c_double), ("up", C.c_int32), ("down", C.c_int32),
                ("in_chunk", C.c_int32), ("max_chunks", C.c_int32),
                ("ntaps_dec", C.c_int32), ("ntaps_af", C.c_int32),
                ("device", C.c_int32), ("reserved", C.c_int32)]


class AgcState(C.Structure):
    _fields_ = [("agc", C.c_float), ("gain", C.c_float), ("maxbuf", C.c_float),
                ("ref", C.c_float), ("err", C.c_float)]


class Out(C.Structure):
    _fields_ = [("am", C.POINTER(C.c_float)), ("iq", C.POINTER(C.c_float)),
                ("cap", C.c_int32), ("n_out", C.c_int32),
                ("am_is_complex", C.c_int32), ("peak_in", C.c_float)]


_vp, _i, _sz, _d, _f, _u32 = C.c_void_p, C.c_int, C.c_size_t, C.c_double, C.c_float, C.c_uint32
_pd, _pf, _pi = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol include/pysdr_hip.h declares
PROTOTYPES = {
    "pysdr_strerror": (C.c_char_p, [_i]),
    "pysdr_last_error": (C.c_char_p, []),
    "pysdr_device_count": (_i, [_pi]),
    "pysdr_version": (_i, []),
    "pysdr_create": (_i, [C.POINTER(Cfg), C.POINTER(_vp)]),
    "pysdr_destroy": (None, [_vp]),
    "pysdr_rx_add": (_i, [_vp, _i, _d, _pd, _pd, _d, _pi]),
    "pysdr_set_lo": (_i, [_vp, _i, _d, _pd]),
    "pysdr_set_dec_taps": (_i, [_vp, _i, _pd, _i]),
    "pysdr_set_mode": (_i, [_vp, _i, _i, _pd, _i, _d]),
    "pysdr_wfm_params": (_i, [_d, _d, _pi, _pi, _pi]),
    "pysdr_set_wfm_taps": (_i, [_vp, _i, _pd, _i, _pd, _i]),
    "pysdr_reset": (_i, [_vp, _i, C.c_uint]),
    "pysdr_agc_get": (_i, [_vp, _i, C.POINTER(AgcState)]),
    "pysdr_pll_stats": (_i, [_vp, _i, _pi, _pi]),
    "pysdr_set_pll_segments": (_i, [_vp, _i]),
    "pysdr_build_flags_hash": (_i, []),
    "pysdr_set_overlap": (_i, [_vp, _i]),
    "pysdr_get_overlap": (_i, [_vp]),
    "pysdr_last_call_overlapped": (_i, [_vp]),
    "pysdr_pll_join_margin": (_i, [_vp, _i, _pi, _pf]),
    "pysdr_pll_linear_starts": (_i, [_vp, _i, _pi]),
    "pysdr_set_agc": (_i, [_vp, _i, _i, _f]),
    "pysdr_set_squelch": (_i, [_vp, _i, _f]),
    "pysdr_squelch_get": (_i, [_vp, _i, _pf, _pi]),
    "pysdr_set_squelch_ratio": (_i, [_vp, _i, _f, _pf, _pf, _i]),
    "pysdr_squelch_ratio_get": (_i, [_vp, _i, _pf, _pf, _pi]),
    "pysdr_process": (_i, [_vp, _pf, _sz, C.POINTER(Out)]),
    "pysdr_process_batch": (_i, [_vp, _vp, _i, _sz, _i]),
    "pysdr_fetch": (_i, [_vp, _i, _pf, _pf, _i, _pi, _pi, _pi, _pf]),
    "pysdr_sync": (_i, [_vp]),
    "pysdr_set_profile": (_i, [_vp, _i]),
    "pysdr_get_elapsed_ms": (_i, [_vp, _i, _i, _pf]),
    "pysdr_set_tile": (_i, [_vp, _i, _i]),
    "pysdr_get_tuning": (_i, [_vp, C.POINTER(C.c_int32)]),
    "pysdr_quad_mixer": (_i, [_i, _pf, _pf, _sz, _u32, _u32, C.POINTER(_u32)]),
    "pysdr_freq_word": (_u32, [_d, _d, _pd]),
    "pysdr_fir_real": (_i, [_i, _pf, _pf, _i, _pf, _sz]),
    "pysdr_spectrum_create": (_i, [_i, _i, _i, _i, _pf, C.POINTER(_vp)]),
    "pysdr_spectrum_destroy": (None, [_vp]),
    "pysdr_spectrum_frame": (_i, [_vp, _pf, _i, _i, _pf, _pi]),
    "pysdr_spectrum_batch": (_i, [_vp, _vp, _i, _sz, _vp]),
    "pysdr_spectrum_sync": (_i, [_vp]),
    "pysdr_spectrum_get_tuning": (_i, [_vp, C.POINTER(C.c_int32)]),
    "pysdr_spectrum_elapsed_ms": (_i, [_vp, _pf]),
    "pysdr_spectrum_order": (_i, [_vp, _vp, _i]),
    "pysdr_ingest_create": (_i, [_vp, _i, C.POINTER(_vp)]),
    "pysdr_ingest_create_batched": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "pysdr_ingest_chunks": (_i, [_vp, _i, _i, _pi, _pi, _pf]),
    "pysdr_ingest_destroy": (None, [_vp]),
    "pysdr_ingest_buffer": (_i, [_vp, _i, C.POINTER(_pf), C.POINTER(C.c_size_t)]),
    "pysdr_ingest_submit": (_i, [_vp, _i, C.c_size_t]),
    "pysdr_ingest_collect": (_i, [_vp, _i, C.POINTER(Out)]),
    "pysdr_waterfall_create": (_i, [_i, _i, _i, C.POINTER(_vp)]),
    "pysdr_waterfall_destroy": (None, [_vp]),
    "pysdr_waterfall_push": (_i, [_vp, _vp, _i, _i]),
    "pysdr_waterfall_roll": (_i, [_vp, _i]),
    "pysdr_waterfall_image": (_i, [_vp, _f, _pf, _pf, _pf]),
    "pysdr_waterfall_image_rows": (_i, [_vp, _f, _i, _pf, _pf, _pf]),
    "pysdr_waterfall_peaks": (_i, [_vp, _pf, _i, C.c_double, _i, C.POINTER(C.c_int32), _i, C.POINTER(C.c_int32)]),
    "pysdr_dev_alloc": (_i, [_i, _sz, C.POINTER(_vp)]),
    "pysdr_dev_free": (_i, [_i, _vp]),
    "pysdr_dev_upload": (_i, [_i, _vp, _vp, _sz]),
    "pysdr_dev_download": (_i, [_i, _vp, _vp, _sz]),
    "pysdr_dev_copy": (_i, [_i, _vp, _vp, _sz]),
    "pysdr_comm_unique_id": (_i, [C.c_char_p]),
    "pysdr_comm_init": (_i, [_vp, C.c_char_p, _i, _i]),
    "pysdr_comm_bcast": (_i, [_vp, _vp, _sz, _i]),
    "pysdr_comm_destroy": (_i, [_vp]),
}

_lib = None


def lib():
    """Load the HIP library (once).  Raises PysdrError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PysdrError(
                f"{LIB_PATH} not found: build it with `python -m pysdr_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:
            L = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
        except OSError as e:
            raise PysdrError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        L = lib()
        msg = L.pysdr_strerror(rc).decode()
        detail = L.pysdr_last_error().decode()
        raise PysdrError(f"{what}: {msg} ({rc}) {detail}")


def device_count():
    n = C.c_int(0)
    rc = lib().pysdr_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def require_gpu():
    if device_count() < 1:
        raise PysdrError("no HIP device visible: pysdr_amd has no CPU path "
                         "(the NumPy oracle under oracle/ is test infrastructure only)")


def as_pd(a):
    return a.ctypes.data_as(_pd)


def as_pf(a):
    return a.ctypes.data_as(_pf)
