"""Round-2 GPU probe: WHERE does the search kernel's two-class timing come from?  One C3 index, one context; the
context's visited bitmaps (512 MB) and then the index's point rows (1.2 GB) are placed at a sweep of offsets inside
ONE big device arena (tuning build hooks idist_tune_set_visited / idist_tune_move_points), everything else fixed.
Also: 4 waves per CU (1024 slots) with many distance rounds in flight.
usage: python scripts/probe_r02_placement.py [out.jsonl]   (GPU box; needs libidist_tune.so)"""
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

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "probe_r02_placement.jsonl")
_capi._singleton = _capi.Lib(os.path.join(ROOT, "instant-distance_amd", "csrc", "libidist_tune.so"))
L = _capi.lib()
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
outs = (torch.empty(nq, 100, dtype=torch.int32, device=dev), torch.empty(nq, 100, dtype=torch.float32, device=dev),
        torch.empty(nq, dtype=torch.int32, device=dev), torch.empty(nq, 3, dtype=torch.int32, device=dev))


def time_ctx(search, reps=3):
    for _ in range(reps + 1):
        h.search_batch_device(search, d_q.data_ptr(), nq, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                              outs[3].data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    search.check_status()
    return round(float(np.median(search.kernel_times_ms(reps))), 3)


# 4 waves per CU with many rounds in flight (is an LDS-resident visited set at 1 wave per SIMD viable?)
for slots in (1024, 1536, 2048):
    for i, nm in ((5, "rif4 qlds occ2"), (8, "rif4 qreg occ2"), (10, "rif8 qlds occ1"), (11, "rif6 qlds occ1"), (12, "rif8 qreg occ1")):
        os.environ["IDIST_TUNE"] = str(i)
        s = ida.Search(slots)
        try:
            emit(what="few fat waves", slots=slots, variant=nm, ms=time_ctx(s, 3))
        except Exception as e:  # noqa: BLE001
            emit(what="few fat waves", slots=slots, variant=nm, error=repr(e))
        del s
os.environ.pop("IDIST_TUNE", None)

GB = 1 << 30
arena_bytes = 24 * GB
arena = torch.empty(arena_bytes, dtype=torch.uint8, device=dev)
base = arena.data_ptr()
emit(what="arena", base=hex(base), bytes=arena_bytes)
info = h.info()
bufs = _capi.DeviceBuffers()
L.check(L.idist_index_device_buffers(h._h, C.byref(bufs)))
emit(what="index buffers", points=hex(bufs.points), zero=hex(bufs.zero), upper=hex(bufs.upper))

s = ida.Search(4096)
emit(what="own allocation", ms=time_ctx(s))
ctx = s._bind(h)
res = []
step = 256 << 20
for k in range(0, 64):
    off = k * step
    L.cdll.idist_tune_set_visited.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    st = L.cdll.idist_tune_set_visited(ctx, C.c_void_p(base + off), 4096)
    assert st == 0
    res.append(time_ctx(s))
emit(what="visited bitmaps at arena offset k*256MB (ms)", ms=res)
# finer: 2 MB steps around the first fast->slow edge, if any
edges = [k for k in range(1, len(res)) if abs(res[k] - res[k - 1]) > 0.4]
if edges:
    k0 = edges[0] - 1
    fine = []
    for j in range(0, 17):
        off = k0 * step + j * (16 << 20)
        L.cdll.idist_tune_set_visited(ctx, C.c_void_p(base + off), 4096)
        fine.append(time_ctx(s))
    emit(what="fine sweep, 16 MB steps from k0", k0=k0, ms=fine)
# now the point rows: visited back at offset 0 of the arena's second half, points swept through the first half
L.cdll.idist_tune_set_visited(ctx, C.c_void_p(base + 16 * GB), 4096)
L.cdll.idist_tune_move_points.argtypes = [C.c_void_p, C.c_void_p]
res = []
for k in range(0, 24):
    st = L.cdll.idist_tune_move_points(h._h, C.c_void_p(base + k * (512 << 20)))
    assert st == 0
    res.append(time_ctx(s))
emit(what="point rows at arena offset k*512MB, visited fixed at +16GB (ms)", ms=res)
