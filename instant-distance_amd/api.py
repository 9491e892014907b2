"""Host-side mirror of instant-distance's public API over the MI355X engine.

Same names, argument meaning and error behaviour as the reference
(/root/reference/instant-distance/src/lib.rs; Python flavour follows
/root/reference/instant-distance-py/src/lib.rs):

    Builder  (core/lib.rs:23-113)     Heuristic (core/lib.rs:115-128)
    Hnsw     (core/lib.rs:194-397)    HnswMap   (core/lib.rs:131-173)
    Search   (core/lib.rs:560-574)    Item / MapItem (core/lib.rs:399-413, 175-191)
    PointId  (core/types.rs:241-253)  -> plain int (u32)

`Point::distance` (core/lib.rs:780-782) is arbitrary user code in the reference and
cannot run on a GPU; here a point is an f32 vector and the distance is one of the two
the reference itself ships: squared L2 (FloatArray, the default) or L2 with sqrt
(the Point of tests/all.rs and examples/colors.rs) — `Builder.metric()`.

All compute goes through the C ABI (include/idist.h); there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import secrets
from dataclasses import dataclass
from typing import Any, Iterator, Sequence

import numpy as np

from . import _capi
from ._capi import INVALID, M, M2, METRIC_L2, METRIC_L2SQ, TIES_DROP, TIES_STRICT

PointId = int


def _lib():
    return _capi.lib()


@dataclass
class Heuristic:
    """core/lib.rs:115-128"""
    extend_candidates: bool = False
    keep_pruned: bool = True


class Builder:
    """Parameters for building the `Hnsw` (core/lib.rs:23-113)."""

    def __init__(self):
        c = _lib().default_config()
        self._ef_search = int(c.ef_search)              # core/lib.rs:104
        self._ef_construction = int(c.ef_construction)  # :105
        self._heuristic: Heuristic | None = Heuristic() # :106
        self._ml = float(c.ml)                          # :107
        self._seed = secrets.randbits(64)               # :108 rand::random()
        self._metric = METRIC_L2SQ
        self._max_batch = 0
        self._device = 0
        self._progress = None
        self._tie_policy = TIES_STRICT
        self._tie_capacity = 0

    @classmethod
    def default(cls) -> "Builder":
        return cls()

    # -- reference setters (core/lib.rs:35-68) --
    def ef_construction(self, ef_construction: int) -> "Builder":
        self._ef_construction = int(ef_construction)
        return self

    def ef_search(self, ef: int) -> "Builder":
        self._ef_search = int(ef)      # does NOT touch ef_construction (code, not the doc comment)
        return self

    def select_heuristic(self, params: Heuristic | None) -> "Builder":
        self._heuristic = params
        return self

    def ml(self, ml: float) -> "Builder":
        self._ml = float(np.float32(ml))
        return self

    def seed(self, seed: int) -> "Builder":
        self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        return self

    def progress(self, bar) -> "Builder":
        """Builder::progress (core/lib.rs:70-75, `indicatif` feature): track the construction.  `bar` is a
        callable `bar(done, total, layer)` — the bar's position, length and the layer named in its message
        (`layer` is None outside the per-layer loop) — called from a watcher thread while the build runs and
        once more when it is finished (`bar.finish()`, :332-334)."""
        self._progress = bar
        return self

    # -- additions of this engine --
    def metric(self, metric: int) -> "Builder":
        """METRIC_L2SQ (FloatArray, py/lib.rs:378-421) or METRIC_L2 (tests/all.rs:93-97)."""
        self._metric = int(metric)
        return self

    def tie_policy(self, policy: int) -> "Builder":
        """TIES_STRICT (default): results are the reference's whatever the number of un-expanded candidates exactly at
        the furthest distance (the tie region grows, then spills to HBM; IdistError(6) only reaches callers of the
        device-pointer API, whose next launch has the room).  TIES_DROP: keep the 64 nearest, go on
        (mass-duplicate data; deterministic, flagged in build_stats().tie_overflow / Search.tie_overflowed())."""
        self._tie_policy = int(policy)
        return self

    def tie_capacity(self, n: int) -> "Builder":
        """Entries of the LDS tie region behind `nearest` (0 = 64, at most 4096).  Strict builds / searches enlarge it
        by themselves and spill to HBM beyond 4096, so this only pre-sizes it for data with masses of exactly equal
        distances (dense integer grids)."""
        self._tie_capacity = int(n)
        return self

    def max_batch(self, k: int) -> "Builder":
        """1 = strictly sequential insertion (deterministic contract); 0 = default."""
        self._max_batch = int(k)
        return self

    def device(self, device: int) -> "Builder":
        self._device = int(device)
        return self

    def into_parts(self):
        """core/lib.rs:87-98"""
        return (self._ef_search, self._ef_construction, self._ml, self._seed)

    def _config(self) -> _capi.Config:
        c = _lib().default_config()
        c.ef_search = self._ef_search
        c.ef_construction = self._ef_construction
        c.ml = self._ml
        c.has_heuristic = 0 if self._heuristic is None else 1
        c.extend_candidates = int(bool(self._heuristic and self._heuristic.extend_candidates))
        c.keep_pruned = int(bool(self._heuristic.keep_pruned)) if self._heuristic else 1
        c.metric = self._metric
        c.max_batch = self._max_batch
        c.tie_policy = self._tie_policy
        c.tie_capacity = self._tie_capacity
        return c

    def build(self, points, values: Sequence[Any]) -> "HnswMap":
        """Builder::build (core/lib.rs:78-80)."""
        return HnswMap._new(points, values, self)

    def build_hnsw(self, points) -> tuple["Hnsw", list[PointId]]:
        """Builder::build_hnsw (core/lib.rs:83-85): (index, original index -> PointId)."""
        return Hnsw._new(points, self)


class _BuildWatch:
    """Arms an idist_progress for the build call of this thread and polls it from a watcher thread."""

    def __init__(self, bar, period: float = 0.05):
        self.bar, self.period, self.h, self._stop, self._thr = bar, period, None, None, None

    def __enter__(self):
        if self.bar is None:
            return self
        import threading

        L = _lib()
        h = C.c_void_p()
        L.check(L.idist_progress_new(C.byref(h)))
        self.h = h
        L.check(L.idist_progress_watch_next_build(h))
        self._stop = threading.Event()

        def poll():
            last = None
            while not self._stop.wait(self.period):
                cur = self._read()
                if cur != last and cur[1]:
                    self.bar(*cur)
                    last = cur

        self._thr = threading.Thread(target=poll, daemon=True)
        self._thr.start()
        return self

    def _read(self):
        d, t, l = C.c_uint64(0), C.c_uint64(0), C.c_int32(-1)
        _lib().idist_progress_get(self.h, C.byref(d), C.byref(t), C.byref(l))
        return int(d.value), int(t.value), (int(l.value) if l.value >= 0 else None)

    def __exit__(self, et, ev, tb):
        if self.h is None:
            return False
        self._stop.set()
        self._thr.join()
        _lib().idist_progress_watch_next_build(None)     # disarm if the build call never consumed it
        if et is None:
            self.bar(*self._read())                       # finish: done == total
        _lib().idist_progress_free(self.h)
        self.h = None
        return False


class Search:
    """Reusable search scratch, `Search::default()` (core/lib.rs:560-574, 767-778).

    After `Hnsw.search(point, search)` it holds that query's results and is an
    iterator over them like the binding's Search (py/lib.rs:177-208)."""

    def __init__(self, slots: int = 0):
        self._slots = slots
        self._ctx = None
        self._owner = None
        self._items: list = []
        self._cur = 0

    def _bind(self, hnsw: "Hnsw"):
        if self._owner is not hnsw:
            self._release()
            ctx = C.c_void_p()
            _lib().check(_lib().idist_search_ctx_new(hnsw._h, self._slots, C.byref(ctx)))
            self._ctx = ctx
            self._owner = hnsw
        return self._ctx

    def _release(self):
        if self._ctx is not None:
            _lib().idist_search_ctx_free(self._ctx)
            self._ctx = None
            self._owner = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def reserve(self, hnsw: "Hnsw", slots: int):
        """Back `slots` query slots now (idist_search_ctx_reserve): later launches up to that width never synchronise
        the device to grow the context."""
        _lib().check(_lib().idist_search_ctx_reserve(self._bind(hnsw), int(slots)))

    def kernel_times_ms(self, last: int = 64) -> np.ndarray:
        """HIP-event durations of the most recent search kernels launched through this Search."""
        out = np.zeros(last, dtype=np.float32)
        n = C.c_uint32(0)
        _lib().check(_lib().idist_search_ctx_kernel_times(self._ctx, _capi.f32p(out), last, C.byref(n)))
        return out[: n.value]

    def tie_overflowed(self) -> bool:
        """TIES_DROP only: did a search through this Search exceed the tie capacity since the last call?"""
        out = C.c_int32(0)
        _lib().check(_lib().idist_search_ctx_tie_overflowed(self._ctx, C.byref(out)))
        return bool(out.value)

    def filter_counts(self, reset: bool = True) -> tuple[int, int]:
        """(candidates the walk's reject filter examined, f32 rows it spared) over this Search's launches since the last reset —
        diagnostics; results never depend on the filter.  Synchronise first."""
        if self._ctx is None:
            return 0, 0
        a, b = C.c_uint64(0), C.c_uint64(0)
        _lib().check(_lib().idist_search_ctx_filter_counts(self._ctx, C.byref(a), C.byref(b), 1 if reset else 0))
        return int(a.value), int(b.value)

    def check_status(self):
        """Raise if a device-side guard tripped during the launches so far (after a stream sync)."""
        _lib().check(_lib().idist_search_ctx_status(self._ctx))

    def __iter__(self):
        return self

    def __next__(self):
        if self._cur >= len(self._items):
            raise StopIteration
        it = self._items[self._cur]
        self._cur += 1
        return it

    def __len__(self):
        return len(self._items)


@dataclass
class Item:
    """core/lib.rs:399-413"""
    distance: float
    pid: PointId
    point: np.ndarray


@dataclass
class MapItem:
    """core/lib.rs:175-191"""
    distance: float
    pid: PointId
    point: np.ndarray
    value: Any


@dataclass
class BatchResult:
    pid: np.ndarray       # [nq, ef_search] uint32, INVALID padded
    distance: np.ndarray  # [nq, ef_search] float32, +inf padded
    count: np.ndarray     # [nq]
    counters: np.ndarray | None  # [nq, 3] {n_dist, n_exp0, n_expU}


def _as_points(points) -> np.ndarray:
    pts = np.ascontiguousarray(points, dtype=np.float32)
    if pts.ndim == 1:
        pts = pts.reshape(0, 1) if pts.size == 0 else pts.reshape(1, -1)
    if pts.ndim != 2:
        raise TypeError("points must be an [n, dim] array of f32")
    return pts


class Hnsw:
    """core/lib.rs:194-397.  Owns the points in PointId order and the device index."""

    def __init__(self, handle, points: np.ndarray, ef_search: int):
        self._h = handle
        self.points = points          # [n, dim] in PointId order (Index<PointId>, core/types.rs:269-275)
        self._ef_search = ef_search

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None:
                _lib().idist_index_free(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def builder() -> Builder:
        return Builder()

    # -- construction --
    @classmethod
    def _new(cls, points, builder: Builder) -> tuple["Hnsw", list[PointId]]:
        """Hnsw::new (core/lib.rs:209-345): shuffle on the host, build on the GPU."""
        pts = _as_points(points)
        n, dim = pts.shape
        L = _lib()
        out_pid = np.zeros(max(n, 1), dtype=np.uint32)
        order = np.zeros(max(n, 1), dtype=np.uint32)
        L.check(L.idist_permutation(C.c_uint64(builder._seed), n, _capi.u32p(out_pid), _capi.u32p(order)))
        ordered = np.ascontiguousarray(pts[order[:n]]) if n else pts.reshape(0, max(dim, 1))
        h = cls.from_ordered_points(ordered, builder)
        return h, [int(x) for x in out_pid[:n]]

    @classmethod
    def from_ordered_points(cls, points_in_pid_order, builder: Builder | None = None) -> "Hnsw":
        """Build from points that already are in PointId order (no shuffle)."""
        builder = builder or Builder()
        pts = _as_points(points_in_pid_order)
        n, dim = pts.shape
        cfg = builder._config()
        h = C.c_void_p()
        L = _lib()
        with _BuildWatch(builder._progress):
            L.check(L.idist_index_build(_capi.f32p(pts), n, max(dim, 1), C.byref(cfg), builder._device, C.byref(h)))
        return cls(h, pts, builder._ef_search)

    @classmethod
    def from_device_points(cls, d_points_ptr: int, n: int, dim: int, builder: Builder | None = None,
                           host_points: np.ndarray | None = None) -> "Hnsw":
        """Build from points already resident in HBM (row-major n x dim f32, PointId order)."""
        builder = builder or Builder()
        if host_points is not None and (tuple(host_points.shape) != (n, dim) or host_points.dtype != np.float32):
            raise ValueError(f"host_points must be the same {n} x {dim} float32 rows that sit on the device, got {host_points.dtype} {host_points.shape}")
        cfg = builder._config()
        h = C.c_void_p()
        L = _lib()
        with _BuildWatch(builder._progress):
            L.check(L.idist_index_build_device(C.c_void_p(d_points_ptr), n, dim, C.byref(cfg), builder._device, C.byref(h)))
        pts = host_points if host_points is not None else np.zeros((n, 0), dtype=np.float32)
        return cls(h, pts, builder._ef_search)

    @classmethod
    def from_parts(cls, points_in_pid_order, zero, layers, builder: Builder | None = None) -> "Hnsw":
        """Adopt the fields of `struct Hnsw` (core/lib.rs:194-199): points, zero, layers."""
        builder = builder or Builder()
        pts = _as_points(points_in_pid_order)
        n, dim = pts.shape
        zero = np.ascontiguousarray(zero, dtype=np.uint32).reshape(n, M2)
        layers = [np.ascontiguousarray(l, dtype=np.uint32).reshape(-1, M) for l in layers]
        ptrs = (C.POINTER(C.c_uint32) * max(len(layers), 1))(*[_capi.u32p(l) for l in layers])
        lens = np.array([l.shape[0] for l in layers] + [0], dtype=np.uint32)
        cfg = builder._config()
        h = C.c_void_p()
        L = _lib()
        L.check(L.idist_index_import(_capi.f32p(pts), n, max(dim, 1), C.byref(cfg), _capi.u32p(zero), ptrs,
                                     _capi.u32p(lens), len(layers), builder._device, C.byref(h)))
        return cls(h, pts, builder._ef_search)

    # -- introspection --
    def info(self) -> _capi.IndexInfo:
        info = _capi.IndexInfo()
        _lib().check(_lib().idist_index_get_info(self._h, C.byref(info)))
        return info

    def into_parts(self):
        """(zero [n,64], layers [[len,32], ...]) copied back from the device."""
        info = self.info()
        zero = np.zeros((info.n, M2), dtype=np.uint32)
        layers = [np.zeros((info.layer_len[l], M), dtype=np.uint32) for l in range(info.n_upper)]
        ptrs = (C.POINTER(C.c_uint32) * max(len(layers), 1))(*[_capi.u32p(l) for l in layers])
        _lib().check(_lib().idist_index_export(self._h, _capi.u32p(zero), ptrs))
        return zero, layers

    def build_stats(self) -> _capi.BuildStats:
        st = _capi.BuildStats()
        _lib().check(_lib().idist_index_build_stats(self._h, C.byref(st)))
        return st

    def set_ef_search(self, ef: int):
        _lib().check(_lib().idist_index_set_ef_search(self._h, int(ef)))
        self._ef_search = int(ef)

    def __len__(self):
        return self.points.shape[0]

    def __getitem__(self, pid: PointId) -> np.ndarray:
        return self.points[pid]

    def iter(self) -> Iterator[tuple[PointId, np.ndarray]]:
        """core/lib.rs:386-391"""
        return ((i, p) for i, p in enumerate(self.points))

    # -- search --
    def search_batch(self, queries, search: Search, counters: bool = False) -> BatchResult:
        """Hnsw::search (core/lib.rs:352-383) for many queries in one launch."""
        q = _as_points(queries)
        dim = self.info().dim     # (the host copy of the points is optional: device-built / replicated indexes)
        if q.shape[0] and len(self) and q.shape[1] != dim:
            raise TypeError(f"query dim {q.shape[1]} != index dim {dim}")
        nq = q.shape[0]
        ef = self._ef_search
        pid = np.full((nq, ef), INVALID, dtype=np.uint32)
        dist = np.full((nq, ef), np.inf, dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        ctr = np.zeros((nq, 3), dtype=np.uint32) if counters else None
        if nq:
            ctx = search._bind(self)
            L = _lib()
            L.check(L.idist_search_batch(self._h, ctx, _capi.f32p(q), nq, _capi.u32p(pid), _capi.f32p(dist),
                                         _capi.u32p(cnt), _capi.u32p(ctr) if counters else None))
        return BatchResult(pid, dist, cnt, ctr)

    def search_batch_device(self, search: Search, d_queries: int, nq: int, d_pid: int, d_dist: int, d_count: int,
                            d_counters: int = 0, stream: int = 0):
        """Device-pointer variant: inputs/outputs stay in HBM, enqueued on `stream` without a sync."""
        ctx = search._bind(self)
        L = _lib()
        L.check(L.idist_search_batch_device(self._h, ctx, C.c_void_p(d_queries), nq, C.c_void_p(d_pid), C.c_void_p(d_dist),
                                            C.c_void_p(d_count), C.c_void_p(d_counters) if d_counters else None,
                                            C.c_void_p(stream) if stream else None))

    # -- several GPUs of one node (single process; the multi-process flavour is dist.py) --
    def replicate(self, devices: Sequence[int], rccl: bool = False) -> list["Hnsw"]:
        """One copy of this index per entry of `devices`, device to device over xGMI: peer copies from the root
        (idist_replicate) or, `rccl=True`, one RCCL broadcast per buffer (idist_replicate_rccl; its wall time is left in
        `self.last_replicate_seconds`).  The replicas share this index's host copy of the points (`Item.point` works on
        every replica)."""
        devs = (C.c_int32 * max(len(devices), 1))(*[int(d) for d in devices])
        outs = (C.c_void_p * max(len(devices), 1))()
        if rccl:
            secs = C.c_double(0.0)
            _lib().check(_lib().idist_replicate_rccl(self._h, devs, len(devices), outs, C.byref(secs)))
            self.last_replicate_seconds = secs.value
        else:
            _lib().check(_lib().idist_replicate(self._h, devs, len(devices), outs))
        return [Hnsw(C.c_void_p(outs[i]), self.points, self._ef_search) for i in range(len(devices))]

    @staticmethod
    def search_batch_sharded(replicas: Sequence["Hnsw"], searches: Sequence[Search], queries, counters: bool = False) -> BatchResult:
        """Hnsw::search for a batch block-partitioned over (replica, Search) pairs — one GPU each, one host
        thread each, no collective (idist_search_batch_sharded).  Identical to `search_batch` on one replica."""
        if not replicas or len(replicas) != len(searches):
            raise ValueError("need one Search per replica")
        q = _as_points(queries)
        nq = q.shape[0]
        ef = replicas[0]._ef_search
        pid = np.full((nq, ef), INVALID, dtype=np.uint32)
        dist = np.full((nq, ef), np.inf, dtype=np.float32)
        cnt = np.zeros(nq, dtype=np.uint32)
        ctr = np.zeros((nq, 3), dtype=np.uint32) if counters else None
        k = len(replicas)
        hs = (C.c_void_p * k)(*[r._h for r in replicas])
        cs = (C.c_void_p * k)(*[s._bind(r) for r, s in zip(replicas, searches)])
        L = _lib()
        L.check(L.idist_search_batch_sharded(hs, cs, k, _capi.f32p(q), nq, _capi.u32p(pid), _capi.f32p(dist), _capi.u32p(cnt),
                                             _capi.u32p(ctr) if counters else None))
        return BatchResult(pid, dist, cnt, ctr)

    def search(self, point, search: Search) -> Search:
        """Search the index for the points nearest to `point` (core/lib.rs:352-383).

        Returns `search`, now an iterator over <= ef_search `Item`s, nearest first."""
        r = self.search_batch(np.asarray(point, dtype=np.float32).reshape(1, -1), search)
        c = int(r.count[0])
        search._items = [Item(float(r.distance[0, i]), int(r.pid[0, i]), self.points[int(r.pid[0, i])]) for i in range(c)]
        search._cur = 0
        return search

    def get(self, i: int, search: Search) -> Item | None:
        """core/lib.rs:393-396"""
        return search._items[i] if 0 <= i < len(search._items) else None

    def distances(self, queries, ids) -> np.ndarray:
        """Point::distance over id lists (the gather-L2 kernel on its own)."""
        q = _as_points(queries)
        ids = np.ascontiguousarray(ids, dtype=np.uint32).reshape(q.shape[0], -1)
        out = np.zeros(ids.shape, dtype=np.float32)
        L = _lib()
        L.check(L.idist_distance_batch(self._h, _capi.f32p(q), q.shape[0], _capi.u32p(ids), ids.shape[1], _capi.f32p(out)))
        return out

    def filter_bounds(self, queries, ids) -> np.ndarray:
        """The walk's reject filter over id lists: a lower bound of every distance, from the compact copy of the rows alone
        (0 where there is none) — idist_filter_bound_batch."""
        q = _as_points(queries)
        ids = np.ascontiguousarray(ids, dtype=np.uint32).reshape(q.shape[0], -1)
        out = np.zeros(ids.shape, dtype=np.float32)
        L = _lib()
        L.check(L.idist_filter_bound_batch(self._h, _capi.f32p(q), q.shape[0], _capi.u32p(ids), ids.shape[1], _capi.f32p(out)))
        return out

    def bruteforce(self, queries, k: int):
        """Exact k-NN by exhaustive scan (the check in tests/all.rs:60-67)."""
        q = _as_points(queries)
        pid = np.zeros((q.shape[0], k), dtype=np.uint32)
        dist = np.zeros((q.shape[0], k), dtype=np.float32)
        L = _lib()
        L.check(L.idist_bruteforce(self._h, _capi.f32p(q), q.shape[0], k, _capi.u32p(pid), _capi.f32p(dist)))
        return pid, dist


class HnswMap:
    """core/lib.rs:131-173"""

    def __init__(self, hnsw: Hnsw, values: list):
        self.hnsw = hnsw
        self.values = values

    @classmethod
    def _new(cls, points, values: Sequence[Any], builder: Builder) -> "HnswMap":
        hnsw, ids = Hnsw._new(points, builder)
        # values re-ordered by PointId (core/lib.rs:144-149); too few values is a panic there (:148)
        if len(values) < len(ids):
            raise IndexError("values.len() < points.len() (core/lib.rs:148 panics)")
        new = [None] * len(ids)
        for src, pid in enumerate(ids):
            new[pid] = values[src]
        return cls(hnsw, new)

    def search(self, point, search: Search) -> Search:
        """core/lib.rs:154-162"""
        self.hnsw.search(point, search)
        search._items = [MapItem(it.distance, it.pid, it.point, self.values[it.pid]) for it in search._items]
        return search

    def iter(self):
        return self.hnsw.iter()

    def get(self, i: int, search: Search):
        return search._items[i] if 0 <= i < len(search._items) else None
