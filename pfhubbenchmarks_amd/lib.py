"""ctypes binding of libpfhip.so (include/pfhip.h).  The library is the product: there is NO fallback path --
if it is missing or fails to load, importing this module's `load()` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpfhip.so")

PF_OK, PF_ERR_INVALID, PF_ERR_UNSUPPORTED, PF_ERR_HIP, PF_ERR_STATE, PF_ERR_NOMEM = 0, -1, -2, -3, -4, -5
PF_BC_PERIODIC, PF_BC_MIRROR = 0, 1
PF_SCHEME_FD_EXPLICIT, PF_SCHEME_SPECTRAL_SI, PF_SCHEME_FEM_BE = 0, 1, 2
PF_MODEL_BM1, PF_MODEL_BM6, PF_MODEL_BM2, PF_MODEL_BM3 = 1, 6, 2, 3
PF_FIELD_C, PF_FIELD_MU, PF_FIELD_PHI, PF_FIELD_ETA1, PF_FIELD_U = 0, 1, 2, 3, 7
PF_KERNEL_AUTO, PF_KERNEL_FUSED, PF_KERNEL_TWOPASS = 0, 1, 2
PF_FLAG_BM6_ELIMINATE_PHI = 1
PF_FLAG_WIDE_HALO = 2
PF_FLAG_FEM_ALWAYS_PIVOT = 4
PF_STAT_FEM_ATTEMPTS, PF_STAT_FEM_NPVT_LEVELS = 0, 1


class PfConfig(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32), ("dim", C.c_int32), ("n", C.c_int32 * 3), ("bc", C.c_int32),
        ("scheme", C.c_int32), ("model", C.c_int32), ("kernel", C.c_int32), ("device", C.c_int32),
        ("nranks", C.c_int32), ("rank", C.c_int32), ("force_slab", C.c_int32),
        ("h", C.c_double),
        ("rho_s", C.c_double), ("c_alpha", C.c_double), ("c_beta", C.c_double), ("kappa", C.c_double),
        ("M", C.c_double), ("k", C.c_double), ("eps_r", C.c_double),
        ("stream", C.c_void_p), ("ext_c", C.c_void_p * 2), ("ext_a2a", C.c_void_p * 2), ("ext_phi", C.c_void_p),
        ("flags", C.c_int32), ("max_newton", C.c_int32), ("model_params", C.c_double * 8),
    ]


class PfStepInfo(C.Structure):
    _fields_ = [("ok", C.c_int32), ("nsteps", C.c_int32), ("cmin", C.c_double), ("cmax", C.c_double),
                ("iters", C.c_int32), ("reserved0", C.c_int32)]


class PfHaloLayout(C.Structure):
    _fields_ = [
        ("base", C.c_void_p), ("plane_elems", C.c_int64), ("ghost", C.c_int32), ("n_local", C.c_int32),
        ("send_lo_off", C.c_int64), ("send_hi_off", C.c_int64), ("recv_lo_off", C.c_int64),
        ("recv_hi_off", C.c_int64), ("rank_lo", C.c_int32), ("rank_hi", C.c_int32), ("cur_index", C.c_int32),
        ("needs_exchange", C.c_int32),
    ]


PF_DIST_DONE, PF_DIST_ALLTOALL, PF_DIST_HALO = 0, 1, 2
PF_DIST_OP_STEP, PF_DIST_OP_REFRESH = 1, 2


class PfDistRequest(C.Structure):
    _fields_ = [("kind", C.c_int32), ("n_halo", C.c_int32), ("src", C.c_void_p), ("dst", C.c_void_p),
                ("doubles_per_peer", C.c_int64), ("halo_base", C.c_void_p * 2)]


class PfkChParams(C.Structure):
    _fields_ = [("c_alpha", C.c_double), ("c_beta", C.c_double), ("two_rho", C.c_double),
                ("kappa_over_h2", C.c_double), ("dtM_over_h2", C.c_double), ("k_phi", C.c_double),
                ("gq", C.c_double), ("cbar", C.c_double)]


# every symbol include/pfhip.h declares: name -> (restype, argtypes)
_H = C.c_void_p
_D = C.POINTER(C.c_double)
SYMBOLS = {
    "pf_version": (C.c_int, []),
    "pf_last_error": (C.c_char_p, [_H]),
    "pf_device_count": (C.c_int, []),
    "pf_status_string": (C.c_char_p, [_H]),
    "pf_config_default": (C.c_int, [C.POINTER(PfConfig), C.c_int, C.c_int, C.c_double]),
    "pf_config_model_defaults": (C.c_int, [C.POINTER(PfConfig), C.c_int]),
    "pf_set_ic_bm2": (C.c_int, [_H, C.c_double, C.c_double, C.c_double, C.c_double]),
    "pf_set_ic_bm3": (C.c_int, [_H, C.c_double, C.c_double, C.c_double, C.c_double]),
    "pf_slab_partition": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pf_field_elems_with_ghosts": (C.c_int64, [C.POINTER(PfConfig)]),
    "pf_field_elems": (C.c_int64, [C.POINTER(PfConfig)]),
    "pf_ext_buffer_offset": (C.c_int64, [C.POINTER(PfConfig), C.c_int]),
    "pf_a2a_buffer_doubles": (C.c_int64, [C.POINTER(PfConfig)]),
    "pf_dist_begin": (C.c_int, [_H, C.c_int, C.c_double]),
    "pf_dist_advance": (C.c_int, [_H, C.POINTER(PfDistRequest)]),
    "pf_create": (C.c_int, [C.POINTER(PfConfig), C.POINTER(_H)]),
    "pf_destroy": (C.c_int, [_H]),
    "pf_set_ic_bm1": (C.c_int, [_H, C.c_double, C.c_double]),
    "pf_set_ic_bm6": (C.c_int, [_H, C.c_double, C.c_double]),
    "pf_set_field": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_size_t]),
    "pf_get_field": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_size_t]),
    "pf_step": (C.c_int, [_H, C.c_double, C.c_int, C.POINTER(PfStepInfo)]),
    "pf_rollback": (C.c_int, [_H]),
    "pf_set_mean_c": (C.c_int, [_H, C.c_double]),
    "pf_sync": (C.c_int, [_H]),
    "pf_halo_layout_get": (C.c_int, [_H, C.POINTER(PfHaloLayout)]),
    "pf_field_halo_layout": (C.c_int, [_H, C.c_int, C.POINTER(PfHaloLayout)]),
    "pf_step_begin": (C.c_int, [_H, C.c_double]),
    "pf_step_finish": (C.c_int, [_H]),
    "pf_set_strip_stream": (C.c_int, [_H, C.c_void_p]),
    "pf_step_slab_fused": (C.c_int, [_H, C.c_double, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pf_diagnostics": (C.c_int, [_H, _D]),
    "pf_diagnostics_local": (C.c_int, [_H, _D]),
    "pf_get_stat": (C.c_int, [_H, C.c_int, C.POINTER(C.c_int64)]),
    "pf_timing_enable": (C.c_int, [_H, C.c_int]),
    "pf_timing_read": (C.c_int, [_H, _D, C.POINTER(C.c_int64)]),
    "pf_timing_samples": (C.c_int, [_H, _D, _D, C.c_int64, C.POINTER(C.c_int64)]),
    "pfk_clock_probe": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "pfk_flags_alloc": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "pfk_flags_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pfk_ipc_export": (C.c_int, [C.c_void_p, C.c_char_p]),
    "pfk_ipc_import": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "pfk_ipc_close": (C.c_int, [C.c_void_p]),
    "pfk_ch_fd_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.POINTER(PfkChParams), C.c_int, C.c_void_p]),
    "pfk_set_tuning": (C.c_int, [C.c_int, C.c_int]),
    "pfk_stream_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "pf_device_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "pf_device_free": (C.c_int, [C.c_void_p]),
    "pfk_grid_barrier_probe": (C.c_int, [C.c_int, C.c_int, C.c_int, _D]),
    "pfk_xcd_barrier_probe": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _D, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pfk_push_planes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pfk_signal_flag": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "pfk_wait_flag": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
}

_lib = None


class PfhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libpfhip error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load libpfhip.so and bind every declared symbol.  Raises if the HIP extension is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libpfhip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C pfhubbenchmarks_amd/csrc` (expected at %s)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def default_config(dim, n, h):
    lib = load()
    cfg = PfConfig()
    rc = lib.pf_config_default(C.byref(cfg), dim, n, h)
    if rc != 0:
        raise PfhipError(rc, "pf_config_default(dim=%r, n=%r, h=%r)" % (dim, n, h))
    return cfg


def check(rc, handle=None):
    if rc != 0:
        msg = load().pf_last_error(handle)
        raise PfhipError(rc, msg.decode() if msg else "?")
    return rc
