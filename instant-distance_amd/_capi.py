"""ctypes binding of the C ABI declared in include/idist.h.

The library is instant-distance_amd/csrc/libidist.so (HIP, gfx950).  There is no
CPU implementation behind this binding: if the shared object is missing, or no
MI355X is visible, every compute call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libidist.so")

M = 32
M2 = 64
INVALID = 0xFFFFFFFF
MAX_LAYERS = 64
MAX_EF = 4096
METRIC_L2SQ = 0
METRIC_L2 = 1
TIES_STRICT = 0
TIES_DROP = 1

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_BAD_GRAPH, ERR_TIE_OVERFLOW, ERR_INTERNAL = range(8)


class IdistError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"idist status {status}: {message}")
        self.status = status
        self.message = message


class Config(C.Structure):
    _fields_ = [
        ("ef_search", C.c_uint32),
        ("ef_construction", C.c_uint32),
        ("ml", C.c_float),
        ("has_heuristic", C.c_int32),
        ("extend_candidates", C.c_int32),
        ("keep_pruned", C.c_int32),
        ("metric", C.c_int32),
        ("max_batch", C.c_uint32),
        ("tie_policy", C.c_int32),
        ("tie_capacity", C.c_uint32),
    ]


class IndexInfo(C.Structure):
    _fields_ = [
        ("n", C.c_uint32), ("dim", C.c_uint32), ("row_stride", C.c_uint32), ("n_upper", C.c_uint32),
        ("ef_search", C.c_uint32), ("metric", C.c_int32), ("device", C.c_int32),
        ("layer_len", C.c_uint32 * MAX_LAYERS),
        ("tie_capacity", C.c_uint32),
    ]


class DeviceBuffers(C.Structure):
    _fields_ = [
        ("points", C.c_void_p), ("points_bytes", C.c_size_t),
        ("zero", C.c_void_p), ("zero_bytes", C.c_size_t),
        ("upper", C.c_void_p), ("upper_bytes", C.c_size_t),
    ]


class BuildStats(C.Structure):
    _fields_ = [
        ("n_dist", C.c_uint64), ("n_exp0", C.c_uint64), ("n_expU", C.c_uint64),
        ("n_sel_pairs", C.c_uint64), ("n_heur_rows", C.c_uint64), ("n_updates", C.c_uint64),
        ("n_updates_fast", C.c_uint64), ("n_updates_full", C.c_uint64),
        ("n_batches", C.c_uint64), ("seconds", C.c_double), ("tie_overflow", C.c_uint64), ("n_heur_ref", C.c_uint64),
        ("n_filter_examined", C.c_uint64), ("n_filter_rejected", C.c_uint64), ("filter_row_bytes", C.c_uint64),
    ]


# every symbol include/idist.h declares: name -> (restype, argtypes)
_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_vp = C.c_void_p
SYMBOLS = {
    "idist_last_error": (C.c_char_p, []),
    "idist_version": (C.c_char_p, []),
    "idist_device_count": (C.c_int32, [C.POINTER(C.c_int32)]),
    "idist_default_config": (C.c_int32, [C.POINTER(Config)]),
    "idist_layer_sizes": (C.c_int32, [C.c_uint32, C.c_float, _u32p, C.c_uint32, _u32p]),
    "idist_permutation": (C.c_int32, [C.c_uint64, C.c_uint32, _u32p, _u32p]),
    "idist_index_build": (C.c_int32, [_f32p, C.c_uint32, C.c_uint32, C.POINTER(Config), C.c_int32, C.POINTER(_vp)]),
    "idist_index_build_device": (C.c_int32, [_vp, C.c_uint32, C.c_uint32, C.POINTER(Config), C.c_int32, C.POINTER(_vp)]),
    "idist_index_build_stats": (C.c_int32, [_vp, C.POINTER(BuildStats)]),
    "idist_progress_new": (C.c_int32, [C.POINTER(_vp)]),
    "idist_progress_free": (None, [_vp]),
    "idist_progress_watch_next_build": (C.c_int32, [_vp]),
    "idist_progress_get": (C.c_int32, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]),
    "idist_index_import": (C.c_int32, [_f32p, C.c_uint32, C.c_uint32, C.POINTER(Config), _u32p, C.POINTER(_u32p), _u32p,
                                       C.c_uint32, C.c_int32, C.POINTER(_vp)]),
    "idist_index_alloc": (C.c_int32, [C.c_uint32, C.c_uint32, C.POINTER(Config), _u32p, C.c_uint32, C.c_int32, C.POINTER(_vp)]),
    "idist_index_export": (C.c_int32, [_vp, _u32p, C.POINTER(_u32p)]),
    "idist_index_get_info": (C.c_int32, [_vp, C.POINTER(IndexInfo)]),
    "idist_index_device_buffers": (C.c_int32, [_vp, C.POINTER(DeviceBuffers)]),
    "idist_index_set_ef_search": (C.c_int32, [_vp, C.c_uint32]),
    "idist_index_free": (None, [_vp]),
    "idist_search_ctx_new": (C.c_int32, [_vp, C.c_uint32, C.POINTER(_vp)]),
    "idist_search_ctx_free": (None, [_vp]),
    "idist_search_ctx_reserve": (C.c_int32, [_vp, C.c_uint32]),
    "idist_search_batch": (C.c_int32, [_vp, _vp, _f32p, C.c_uint32, _u32p, _f32p, _u32p, _u32p]),
    "idist_search_batch_device": (C.c_int32, [_vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp]),
    "idist_search_ctx_status": (C.c_int32, [_vp]),
    "idist_search_ctx_tie_overflowed": (C.c_int32, [_vp, C.POINTER(C.c_int32)]),
    "idist_search_ctx_filter_counts": (C.c_int32, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int32]),
    "idist_search_ctx_last_kernel_ms": (C.c_int32, [_vp, C.POINTER(C.c_float)]),
    "idist_search_ctx_kernel_times": (C.c_int32, [_vp, _f32p, C.c_uint32, _u32p]),
    "idist_replicate": (C.c_int32, [_vp, C.POINTER(C.c_int32), C.c_uint32, C.POINTER(_vp)]),
    "idist_replicate_rccl": (C.c_int32, [_vp, C.POINTER(C.c_int32), C.c_uint32, C.POINTER(_vp), C.POINTER(C.c_double)]),
    "idist_search_batch_sharded": (C.c_int32, [C.POINTER(_vp), C.POINTER(_vp), C.c_uint32, _f32p, C.c_uint32, _u32p, _f32p,
                                               _u32p, _u32p]),
    "idist_distance_batch": (C.c_int32, [_vp, _f32p, C.c_uint32, _u32p, C.c_uint32, _f32p]),
    "idist_filter_bound_batch": (C.c_int32, [_vp, _f32p, C.c_uint32, _u32p, C.c_uint32, _f32p]),
    "idist_bruteforce": (C.c_int32, [_vp, _f32p, C.c_uint32, C.c_uint32, _u32p, _f32p]),
}


class Lib:
    """A loaded libidist with typed entry points and status checking."""

    def __init__(self, path: str = LIB_PATH):
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with `make -C instant-distance_amd/csrc` "
                "(hipcc --offload-arch=gfx950). instant_distance_amd has no CPU fallback.")
        self.path = path
        # One HIP runtime per process: if torch is already imported, let it bring up ITS runtime first —
        # torch's bundled libamdhip64 refuses to initialise ("No HIP GPUs are available") once another copy
        # has opened the device, while libidist happily shares torch's (same SONAME).
        import sys as _sys
        _torch = _sys.modules.get("torch")
        if _torch is not None:
            try:
                if _torch.cuda.is_available():
                    _torch.cuda.init()
            except Exception:  # noqa: BLE001
                pass
        self.cdll = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(self.cdll, name)   # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)

    def check(self, status: int):
        if status != OK:
            raise IdistError(status, self.idist_last_error().decode(errors="replace"))

    def default_config(self) -> Config:
        c = Config()
        self.check(self.idist_default_config(C.byref(c)))
        return c

    def device_count(self) -> int:
        n = C.c_int32(0)
        self.check(self.idist_device_count(C.byref(n)))
        return n.value


_singleton: Lib | None = None


def lib() -> Lib:
    global _singleton
    if _singleton is None:
        _singleton = Lib()
    return _singleton


def f32p(a: np.ndarray):
    return a.ctypes.data_as(_f32p)


def u32p(a: np.ndarray):
    return a.ctypes.data_as(_u32p)
