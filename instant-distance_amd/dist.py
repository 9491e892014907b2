"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for tests).

HNSW search shards trivially (Hnsw::search takes &self, all mutable state is in Search —
core/lib.rs:352-356): the index is replicated ONCE by broadcast, the query batch is split
into contiguous ranges, every rank writes its own result slab.  No collective in steady
state.  The build does not shard (every insert reads and mutates one graph): it runs on
one GPU and is replicated.
"""
from __future__ import annotations

# NOTE: import torch (and touch torch.cuda) BEFORE the first instant_distance_amd call of the process when
# you intend to use the device views: both libraries must share one HIP runtime (see _capi.Lib).
import ctypes as C

import numpy as np

from . import _capi
from .api import Builder, Hnsw


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block partition of [0, n_items) (the reference's concurrency model:
    callers block-partition queries over their own threads)."""
    lo = (n_items * rank) // world
    hi = (n_items * (rank + 1)) // world
    return lo, hi


class _DevView:
    """Expose a raw device allocation to torch without copying."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_views(hnsw: Hnsw, device):
    """uint8 torch tensors aliasing the index's points / zero / upper device buffers."""
    import torch

    bufs = _capi.DeviceBuffers()
    _capi.lib().check(_capi.lib().idist_index_device_buffers(hnsw._h, C.byref(bufs)))
    out = []
    for ptr, nb in ((bufs.points, bufs.points_bytes), (bufs.zero, bufs.zero_bytes), (bufs.upper, bufs.upper_bytes)):
        out.append(torch.as_tensor(_DevView(ptr, nb), device=device) if nb else None)
    return out


def device_buffer_bytes(hnsw: Hnsw) -> int:
    """Bytes of the three device buffers a replication moves (points + zero layer + upper layers)."""
    bufs = _capi.DeviceBuffers()
    _capi.lib().check(_capi.lib().idist_index_device_buffers(hnsw._h, C.byref(bufs)))
    return int(bufs.points_bytes + bufs.zero_bytes + bufs.upper_bytes)


def _bcast_meta(meta: np.ndarray, src: int):
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(meta)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def replicate_index(hnsw: Hnsw | None, builder: Builder, src: int = 0, chunk_bytes: int = 1 << 30) -> Hnsw:
    """Give every rank a replica of rank `src`'s index.

    nccl: zero-copy — the destination ranks allocate an empty index of the same layer
    structure (idist_index_alloc) and RCCL broadcasts straight into its device buffers in
    <= chunk_bytes pieces (xGMI is point-to-point; big messages keep every link streaming).
    gloo (CPU tests): export -> broadcast host arrays -> import.
    """
    import torch
    import torch.distributed as dist

    rank = dist.get_rank()
    L = _capi.lib()
    meta = np.zeros(8 + _capi.MAX_LAYERS, dtype=np.int64)
    if rank == src:
        info = hnsw.info()
        meta[0], meta[1], meta[2], meta[3] = info.n, info.dim, info.n_upper, info.ef_search
        # does the source hold a host copy of the points (all n rows, f32)?  The host transport broadcasts it as it is.
        hp = hnsw.points
        meta[4] = 1 if (info.n == 0 or (tuple(hp.shape) == (info.n, info.dim) and hp.dtype == np.float32)) else 0
        meta[8:8 + info.n_upper] = list(info.layer_len)[: info.n_upper]
    meta = _bcast_meta(meta, src)
    n, dim, n_upper, ef = int(meta[0]), int(meta[1]), int(meta[2]), int(meta[3])
    layer_len = np.ascontiguousarray(meta[8:8 + n_upper].astype(np.uint32))

    def _all_ok(ok: bool) -> bool:
        """every rank must have finished its LOCAL preparation before the first bulk collective, otherwise a
        rank that failed locally would leave the others blocked inside the broadcast"""
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    if dist.get_backend() == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        views, err = None, None
        try:
            if rank != src:
                cfg = builder._config()
                cfg.ef_search = ef
                h = C.c_void_p()
                L.check(L.idist_index_alloc(n, dim, C.byref(cfg), _capi.u32p(layer_len), n_upper, dev.index, C.byref(h)))
                hnsw = Hnsw(h, np.zeros((n, 0), dtype=np.float32), ef)   # host copy of the points is not replicated
            views = device_views(hnsw, dev)
        except Exception as e:  # noqa: BLE001
            err = e
        if not _all_ok(err is None):
            raise RuntimeError(f"replicate_index: local preparation failed on some rank (this rank: {err!r})")
        for t in views:
            if t is None:
                continue
            for off in range(0, t.numel(), chunk_bytes):
                dist.broadcast(t[off:off + chunk_bytes], src=src)
        torch.cuda.synchronize()
        return hnsw

    # host transport
    if not int(meta[4]):
        # (every rank sees the same flag: all of them stop here, none is left inside a broadcast)
        raise RuntimeError("replicate_index over a host transport needs the source's host copy of the points: build with "
                           "Hnsw.from_ordered_points, or pass host_points= to Hnsw.from_device_points")
    if rank == src:
        zero, layers = hnsw.into_parts()
        pts = np.ascontiguousarray(hnsw.points, dtype=np.float32)
    else:
        zero = np.zeros((n, _capi.M2), dtype=np.uint32)
        layers = [np.zeros((int(k), _capi.M), dtype=np.uint32) for k in layer_len]
        pts = np.zeros((n, dim), dtype=np.float32)
    for arr in [pts, zero] + layers:
        if arr.size:
            dist.broadcast(torch.from_numpy(arr.view(np.uint8).reshape(-1)), src=src)
    if rank == src:
        return hnsw
    return Hnsw.from_parts(pts, zero, layers, builder.ef_search(ef))
