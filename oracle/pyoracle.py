"""ctypes binding of the CPU ORACLE (oracle/libidist_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by the product package
(instant_distance_amd).  See oracle/idist_oracle.h for the reference
citations of every entry point.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libidist_oracle.so")

M = 32
M2 = 64
INVALID = 0xFFFFFFFF
METRIC_L2SQ = 0
METRIC_L2 = 1


class Config(C.Structure):
    _fields_ = [
        ("ef_search", C.c_uint32),
        ("ef_construction", C.c_uint32),
        ("ml", C.c_float),
        ("has_heuristic", C.c_int32),
        ("extend_candidates", C.c_int32),
        ("keep_pruned", C.c_int32),
        ("metric", C.c_int32),
    ]


class Counters(C.Structure):
    _fields_ = [("n_dist", C.c_uint64), ("n_exp0", C.c_uint64),
                ("n_expU", C.c_uint64), ("n_heur", C.c_uint64)]


def build_lib(force: bool = False) -> str:
    """Compile the oracle with its committed Makefile (gcc)."""
    src = os.path.join(_HERE, "idist_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build_lib()
        L = C.CDLL(_SO)
        f32p = C.POINTER(C.c_float)
        u32p = C.POINTER(C.c_uint32)
        L.ido_default_config.argtypes = [C.POINTER(Config)]
        L.ido_distance.restype = C.c_float
        L.ido_distance.argtypes = [f32p, f32p, C.c_uint32, C.c_int]
        L.ido_distance_scalar.restype = C.c_float
        L.ido_distance_scalar.argtypes = [f32p, f32p, C.c_uint32, C.c_int]
        L.ido_layer_sizes.restype = C.c_uint32
        L.ido_layer_sizes.argtypes = [C.c_uint32, C.c_float, u32p, C.c_uint32]
        L.ido_permutation.argtypes = [C.c_uint64, C.c_uint32, u32p, u32p]
        L.ido_build.restype = C.c_void_p
        L.ido_build.argtypes = [f32p, C.c_uint32, C.c_uint32, C.POINTER(Config), C.c_int,
                                C.POINTER(Counters)]
        L.ido_import.restype = C.c_void_p
        L.ido_import.argtypes = [f32p, C.c_uint32, C.c_uint32, C.POINTER(Config), u32p,
                                 C.POINTER(u32p), u32p, C.c_uint32]
        L.ido_import_borrowed.restype = C.c_void_p
        L.ido_import_borrowed.argtypes = L.ido_import.argtypes
        L.ido_free.argtypes = [C.c_void_p]
        for name in ("ido_n", "ido_dim", "ido_n_upper"):
            getattr(L, name).restype = C.c_uint32
            getattr(L, name).argtypes = [C.c_void_p]
        L.ido_layer_len.restype = C.c_uint32
        L.ido_layer_len.argtypes = [C.c_void_p, C.c_uint32]
        L.ido_zero.restype = u32p
        L.ido_zero.argtypes = [C.c_void_p]
        L.ido_layer.restype = u32p
        L.ido_layer.argtypes = [C.c_void_p, C.c_uint32]
        L.ido_points.restype = f32p
        L.ido_points.argtypes = [C.c_void_p]
        L.ido_set_ef_search.argtypes = [C.c_void_p, C.c_uint32]
        L.ido_search_new.restype = C.c_void_p
        L.ido_search_free.argtypes = [C.c_void_p]
        L.ido_search_one.restype = C.c_uint32
        L.ido_search_one.argtypes = [C.c_void_p, C.c_void_p, f32p, u32p, f32p, C.POINTER(Counters)]
        L.ido_search_batch.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_int, u32p, f32p, u32p, u32p]
        L.ido_bruteforce.argtypes = [f32p, C.c_uint32, C.c_uint32, C.c_int, f32p, C.c_uint32,
                                     C.c_uint32, C.c_int, u32p, f32p]
        _lib = L
    return _lib


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def default_config(**kw) -> Config:
    c = Config()
    lib().ido_default_config(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def distance(a, b, metric=METRIC_L2SQ, scalar=False) -> np.float32:
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    fn = lib().ido_distance_scalar if scalar else lib().ido_distance
    return np.float32(fn(_f32(a), _f32(b), a.size, metric))


def layer_sizes(n: int, ml: float | None = None) -> list[int]:
    """cum[l] = number of nodes present on layer l (cum[0] == n)."""
    if ml is None:
        ml = default_config().ml
    out = np.zeros(64, dtype=np.uint32)
    k = lib().ido_layer_sizes(n, C.c_float(ml), _u32(out), 64)
    return [int(x) for x in out[:k]]


def permutation(seed: int, n: int):
    """(out_pid[orig] -> pid, order[pid] -> orig). rand restatement: PARITY UNPINNED."""
    out = np.zeros(max(n, 1), dtype=np.uint32)
    order = np.zeros(max(n, 1), dtype=np.uint32)
    lib().ido_permutation(C.c_uint64(seed), n, _u32(out), _u32(order))
    return out[:n], order[:n]


@dataclass
class SearchResult:
    pid: np.ndarray      # [nq, ef] uint32 (INVALID padded)
    dist: np.ndarray     # [nq, ef] float32
    count: np.ndarray    # [nq] uint32
    counters: np.ndarray  # [nq, 3] uint32 {n_dist, n_exp0, n_expU}


class Index:
    """Owns an ido_index*."""

    def __init__(self, handle, cfg: Config):
        self._h = handle
        self.cfg = cfg

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ido_free(self._h)
            self._h = None

    @classmethod
    def build(cls, points, cfg: Config | None = None, threads: int = 1):
        cfg = cfg or default_config()
        pts = np.ascontiguousarray(points, dtype=np.float32)
        n, dim = pts.shape
        ctr = Counters()
        h = lib().ido_build(_f32(pts), n, dim, C.byref(cfg), threads, C.byref(ctr))
        ix = cls(h, cfg)
        ix.build_counters = ctr
        return ix

    @classmethod
    def from_arrays(cls, points, zero, layers, cfg: Config | None = None, borrow: bool = False):
        """borrow=True: the index reads `points` / `zero` in place (they must be C-contiguous f32 / u32 and are kept
        alive by the returned object) — for indexes whose points would not fit in host memory twice."""
        cfg = cfg or default_config()
        pts = np.ascontiguousarray(points, dtype=np.float32)
        n, dim = pts.shape
        zero = np.ascontiguousarray(zero, dtype=np.uint32).reshape(n, M2)
        layers = [np.ascontiguousarray(l, dtype=np.uint32).reshape(-1, M) for l in layers]
        ptrs = (C.POINTER(C.c_uint32) * max(len(layers), 1))(*[_u32(l) for l in layers])
        lens = np.array([l.shape[0] for l in layers] + [0], dtype=np.uint32)
        fn = lib().ido_import_borrowed if borrow else lib().ido_import
        h = fn(_f32(pts), n, dim, C.byref(cfg), _u32(zero), ptrs, _u32(lens), len(layers))
        ix = cls(h, cfg)
        if borrow:
            ix._keep = (pts, zero)
        return ix

    @property
    def n(self):
        return lib().ido_n(self._h)

    @property
    def dim(self):
        return lib().ido_dim(self._h)

    @property
    def points(self):
        n, d = self.n, self.dim
        if n == 0:
            return np.zeros((0, d), dtype=np.float32)
        return np.ctypeslib.as_array(lib().ido_points(self._h), shape=(n, d)).copy()

    @property
    def zero(self):
        n = self.n
        if n == 0:
            return np.zeros((0, M2), dtype=np.uint32)
        return np.ctypeslib.as_array(lib().ido_zero(self._h), shape=(n, M2)).copy()

    @property
    def layers(self):
        out = []
        for l in range(1, lib().ido_n_upper(self._h) + 1):
            ln = lib().ido_layer_len(self._h, l)
            out.append(np.ctypeslib.as_array(lib().ido_layer(self._h, l), shape=(ln, M)).copy())
        return out

    def set_ef_search(self, ef: int):
        self.cfg.ef_search = ef
        lib().ido_set_ef_search(self._h, ef)

    def search(self, queries, threads: int = 1) -> SearchResult:
        q = np.ascontiguousarray(queries, dtype=np.float32)
        if q.ndim == 1:
            q = q[None, :]
        nq = q.shape[0]
        ef = max(int(self.cfg.ef_search), 1)
        pid = np.full((nq, ef), INVALID, dtype=np.uint32)
        dist = np.full((nq, ef), np.inf, dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        ctr = np.zeros((nq, 3), dtype=np.uint32)
        if nq:
            lib().ido_search_batch(self._h, _f32(q), nq, threads, _u32(pid), _f32(dist), _u32(cnt), _u32(ctr))
        if self.cfg.ef_search == 0:
            cnt[:] = 0
        return SearchResult(pid, dist, cnt, ctr)


def bruteforce(points, queries, k, metric=METRIC_L2SQ, threads=1):
    pts = np.ascontiguousarray(points, dtype=np.float32)
    q = np.ascontiguousarray(queries, dtype=np.float32)
    if q.ndim == 1:
        q = q[None, :]
    n, dim = pts.shape
    nq = q.shape[0]
    pid = np.zeros((nq, k), dtype=np.uint32)
    dist = np.zeros((nq, k), dtype=np.float32)
    lib().ido_bruteforce(_f32(pts), n, dim, metric, _f32(q), nq, k, threads, _u32(pid), _f32(dist))
    return pid, dist
