"""Drop-in for the reference's Python wheel (`import instant_distance`):
/root/reference/instant-distance-py/src/lib.rs, classes Config, Heuristic, Hnsw, HnswMap,
Search, Neighbor (py/lib.rs:18-357), backed by the MI355X engine.

    import instant_distance_amd.pybinding as instant_distance
    config = instant_distance.Config()
    (hnsw, ids) = instant_distance.Hnsw.build(points, config)
    search = instant_distance.Search()
    hnsw.search(p, search)
    for candidate in search: ...

Same semantics as the binding: points are lists of <= 300 floats, zero-padded to 300
(`point array too long` otherwise, py/lib.rs:363-376); the distance is the squared L2 of
FloatArray (py/lib.rs:378-421); `load`/`dump` speak the binding's on-disk format:
bincode 1.3 (fixint, little endian) of the serde derives — see `_dump_hnsw`.

PARITY NOTE: the reference has no serialization test and ships no sample file, and bincode /
serde are not part of /root/reference, so the wire format is restated from the serde derives
(core/lib.rs:130,193; core/types.rs:61,81-85,239; py/lib.rs:359-361,423-426) and is
"parity unpinned" until a file written by the real wheel is available.
"""
from __future__ import annotations

import struct
from typing import Any

import numpy as np

from . import _capi
from . import api as _api

DIMENSIONS = 300   # py/lib.rs:448


class Heuristic:
    """py/lib.rs:276-324"""

    def __init__(self):
        self.extend_candidates = False
        self.keep_pruned = True


class Config:
    """py/lib.rs:216-255"""

    def __init__(self):
        b = _api.Builder()
        self.ef_search, self.ef_construction, self.ml, self.seed = b.into_parts()
        self.heuristic: Heuristic | None = Heuristic()

    def _builder(self) -> _api.Builder:   # From<&Config> for Builder, py/lib.rs:257-274
        h = None if self.heuristic is None else _api.Heuristic(self.heuristic.extend_candidates, self.heuristic.keep_pruned)
        return (_api.Builder().ef_search(self.ef_search).ef_construction(self.ef_construction).ml(self.ml)
                .seed(self.seed).select_heuristic(h))


class Neighbor:
    """py/lib.rs:326-357"""

    def __init__(self, distance: float, pid: int, value: Any = None):
        self.distance, self.pid, self.value = distance, pid, value

    def __repr__(self):
        if self.value is not None:
            return f"instant_distance.Neighbor(distance={self.distance}, pid={self.pid}, value={self.value!r})"
        return f"instant_distance.Item(distance={self.distance}, pid={self.pid})"


def _float_array(point) -> np.ndarray:
    """TryFrom<&PyAny> for FloatArray, py/lib.rs:363-376"""
    out = np.zeros(DIMENSIONS, dtype=np.float32)
    vals = list(point)
    if len(vals) > DIMENSIONS:
        raise TypeError("point array too long")
    out[: len(vals)] = np.asarray(vals, dtype=np.float32)
    return out


def _float_arrays(points) -> np.ndarray:
    if isinstance(points, np.ndarray) and points.ndim == 2 and points.shape[1] <= DIMENSIONS:
        out = np.zeros((points.shape[0], DIMENSIONS), dtype=np.float32)
        out[:, : points.shape[1]] = points
        return out
    return np.stack([_float_array(p) for p in points]) if len(points) else np.zeros((0, DIMENSIONS), np.float32)


class Search:
    """py/lib.rs:177-214: search buffer and result iterator"""

    def __init__(self):
        self._inner = _api.Search()
        self._cur = None

    def __iter__(self):
        return self

    def __next__(self) -> Neighbor:
        if self._cur is None:
            raise StopIteration
        index, idx = self._cur
        item = index._get(idx, self._inner)
        if item is None:
            self._cur = None
            raise StopIteration
        self._cur = (index, idx + 1)
        return item


# ---- bincode 1.3 (default options: fixint, little endian) of the serde derives ----------------
def _dump_hnsw(f, h: _api.Hnsw):
    """struct Hnsw { ef_search: usize, points: Vec<FloatArray>, zero: Vec<ZeroNode>, layers: Vec<Vec<UpperNode>> }
    (core/lib.rs:193-199).  usize -> u64; Vec -> u64 length + elements; FloatArray / ZeroNode are BigArray
    newtypes -> their elements with no length prefix; PointId is a u32 newtype."""
    zero, layers = h.into_parts()
    pts = np.ascontiguousarray(h.points, dtype="<f4")
    f.write(struct.pack("<Q", h._ef_search))
    f.write(struct.pack("<Q", pts.shape[0]))
    f.write(pts.tobytes())
    f.write(struct.pack("<Q", zero.shape[0]))
    f.write(np.ascontiguousarray(zero, dtype="<u4").tobytes())
    f.write(struct.pack("<Q", len(layers)))
    for l in layers:
        f.write(struct.pack("<Q", l.shape[0]))
        f.write(np.ascontiguousarray(l, dtype="<u4").tobytes())


def _read(f, n):
    b = f.read(n)
    if len(b) != n:
        raise ValueError("deserialization error: unexpected end of file")
    return b


def _load_hnsw(f) -> _api.Hnsw:
    ef = struct.unpack("<Q", _read(f, 8))[0]
    n = struct.unpack("<Q", _read(f, 8))[0]
    pts = np.frombuffer(_read(f, n * DIMENSIONS * 4), dtype="<f4").reshape(n, DIMENSIONS).astype(np.float32)
    nz = struct.unpack("<Q", _read(f, 8))[0]
    if nz != n:
        raise ValueError(f"deserialization error: {nz} zero nodes for {n} points")
    zero = np.frombuffer(_read(f, nz * _capi.M2 * 4), dtype="<u4").reshape(nz, _capi.M2).astype(np.uint32)
    nl = struct.unpack("<Q", _read(f, 8))[0]
    layers = []
    for _ in range(nl):
        ln = struct.unpack("<Q", _read(f, 8))[0]
        layers.append(np.frombuffer(_read(f, ln * _capi.M * 4), dtype="<u4").reshape(ln, _capi.M).astype(np.uint32))
    return _api.Hnsw.from_parts(pts, zero, layers, _api.Builder().ef_search(int(ef)))


class Hnsw:
    """py/lib.rs:100-175.  300-element f32 vectors, squared Euclidean distance."""

    def __init__(self, inner: _api.Hnsw):
        self._inner = inner

    @staticmethod
    def build(input, config: Config):
        inner, ids = config._builder().build_hnsw(_float_arrays(input))
        return Hnsw(inner), ids

    @staticmethod
    def load(fname: str) -> "Hnsw":
        with open(fname, "rb", buffering=32 * 1024 * 1024) as f:
            return Hnsw(_load_hnsw(f))

    def dump(self, fname: str) -> None:
        with open(fname, "wb", buffering=32 * 1024 * 1024) as f:
            _dump_hnsw(f, self._inner)

    def search(self, point, search: Search) -> None:
        self._inner.search(_float_array(point), search._inner)
        search._cur = (self, 0)

    def _get(self, idx: int, inner: _api.Search):
        it = self._inner.get(idx, inner)
        return None if it is None else Neighbor(it.distance, it.pid, None)


class HnswMap:
    """py/lib.rs:30-98.  Values are strings (enum MapValue::String, py/lib.rs:423-446)."""

    def __init__(self, inner: _api.HnswMap):
        self._inner = inner

    @staticmethod
    def build(points, values, config: Config) -> "HnswMap":
        vals = []
        for v in values:
            if not isinstance(v, str):
                raise TypeError("values must be str (MapValue::String)")
            vals.append(v)
        return HnswMap(config._builder().build(_float_arrays(points), vals))

    @staticmethod
    def load(fname: str) -> "HnswMap":
        with open(fname, "rb", buffering=32 * 1024 * 1024) as f:
            hnsw = _load_hnsw(f)
            n = struct.unpack("<Q", _read(f, 8))[0]
            values = []
            for _ in range(n):
                variant = struct.unpack("<I", _read(f, 4))[0]      # enum MapValue: u32 variant index
                if variant != 0:
                    raise ValueError(f"deserialization error: unknown MapValue variant {variant}")
                ln = struct.unpack("<Q", _read(f, 8))[0]
                values.append(_read(f, ln).decode("utf-8"))
        return HnswMap(_api.HnswMap(hnsw, values))

    def dump(self, fname: str) -> None:
        with open(fname, "wb", buffering=32 * 1024 * 1024) as f:
            _dump_hnsw(f, self._inner.hnsw)                          # struct HnswMap { hnsw, values }, core/lib.rs:130-134
            f.write(struct.pack("<Q", len(self._inner.values)))
            for v in self._inner.values:
                b = v.encode("utf-8")
                f.write(struct.pack("<IQ", 0, len(b)))
                f.write(b)

    def search(self, point, search: Search) -> None:
        self._inner.search(_float_array(point), search._inner)
        search._cur = (self, 0)

    def _get(self, idx: int, inner: _api.Search):
        it = self._inner.get(idx, inner)
        return None if it is None else Neighbor(it.distance, it.pid, it.value)
