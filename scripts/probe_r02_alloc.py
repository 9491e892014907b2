"""Round-2 GPU probe: which ALLOCATION carries the two-class timing of the search kernel?  One C3 index, one context.
Each of the index's buffers (point rows 1.2 GB, zero layer 256 MB) is moved in turn into a series of fresh device
allocations (the others fixed), the search launch and a plain random-row gather over the same allocation are timed.
usage: python scripts/probe_r02_alloc.py [out.jsonl]   (GPU box; needs libidist_tune.so)"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import instant_distance_amd as ida  # noqa: E402
from instant_distance_amd import _capi  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "probe_r02_alloc.jsonl")
_capi._singleton = _capi.Lib(os.path.join(ROOT, "instant-distance_amd", "csrc", "libidist_tune.so"))
L = _capi.lib()
L.cdll.idist_tune_move_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
L.cdll.idist_tune_set_visited.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
fo = open(out_path, "a")


def emit(**kw):
    print(json.dumps(kw), flush=True)
    fo.write(json.dumps(kw) + "\n")
    fo.flush()


dev = torch.device("cuda", 0)
n, dim, nq = 1_000_000, 300, 10_000
d_pts = bench.synth(torch, n, dim, 123456789, dev)
d_q = bench.synth(torch, nq, dim, 123456790, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
del d_pts
outs = (torch.empty(nq, 100, dtype=torch.int32, device=dev), torch.empty(nq, 100, dtype=torch.float32, device=dev),
        torch.empty(nq, dtype=torch.int32, device=dev), torch.empty(nq, 3, dtype=torch.int32, device=dev))
s = ida.Search(4096)


def t_search(reps=3):
    for _ in range(reps + 1):
        h.search_batch_device(s, d_q.data_ptr(), nq, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                              outs[3].data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    s.check_status()
    return round(float(np.median(s.kernel_times_ms(reps))), 3)


perm = torch.randperm(n, device=dev)
gout = torch.empty(n, 304, dtype=torch.float32, device=dev)


class View:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def t_gather(ptr):
    rows = torch.as_tensor(View(ptr, n * 304 * 4), device=dev).view(n, 304)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(4):
        e0.record()
        torch.index_select(rows, 0, perm, out=gout)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return round(best, 4)


bufs = _capi.DeviceBuffers()
L.check(L.idist_index_device_buffers(h._h, C.byref(bufs)))
emit(what="as built", points=hex(bufs.points), zero=hex(bufs.zero), upper=hex(bufs.upper), search_ms=t_search(), gather_ms=t_gather(bufs.points))
orig_points, orig_zero = bufs.points, bufs.zero
keep = []
# ten fresh contexts first (visited allocations), index untouched
tens = []
for i in range(6):
    s2 = ida.Search(4096)
    s_old, s = s, s2
    tens.append(t_search())
    keep.append(s_old)
emit(what="six fresh contexts, index as built", ms=tens)
for i in range(10):
    t = torch.empty(n * 304 * 4, dtype=torch.uint8, device=dev)
    keep.append(t)
    assert L.cdll.idist_tune_move_buffer(h._h, 0, C.c_void_p(t.data_ptr()), C.c_void_p(orig_points)) == 0
    emit(what="points in fresh allocation", i=i, ptr=hex(t.data_ptr()), search_ms=t_search(), gather_ms=t_gather(t.data_ptr()))
assert L.cdll.idist_tune_move_buffer(h._h, 0, C.c_void_p(keep[-1].data_ptr()), C.c_void_p(orig_points)) == 0
for i in range(8):
    t = torch.empty(n * 256, dtype=torch.uint8, device=dev)
    keep.append(t)
    assert L.cdll.idist_tune_move_buffer(h._h, 1, C.c_void_p(t.data_ptr()), C.c_void_p(orig_zero)) == 0
    emit(what="zero layer in fresh allocation", i=i, ptr=hex(t.data_ptr()), search_ms=t_search())
# time series on one fixed configuration: does the level drift with time?
series = [t_search(1) for _ in range(40)]
emit(what="40 consecutive single-launch timings, fixed configuration", ms=series)
