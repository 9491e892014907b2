"""Run under `rocprofv3 --pmc <counters> -d <dir> -o x -- python scripts/probe_r02_alloc_pmc.py`: the C3 search launch
with the point rows in a series of fresh allocations (fast and slow class), 2 launches each; prints the HIP-event time
of every launch in dispatch order so the per-dispatch counters can be matched to the class.  (GPU box, tuning build.)"""
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

_capi._singleton = _capi.Lib(os.path.join(ROOT, "instant-distance_amd", "csrc", "libidist_tune.so"))
L = _capi.lib()
L.cdll.idist_tune_move_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
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
bufs = _capi.DeviceBuffers()
L.check(L.idist_index_device_buffers(h._h, C.byref(bufs)))
orig = bufs.points
seq = []


def launches(tag, k=2):
    for _ in range(k):
        h.search_batch_device(s, d_q.data_ptr(), nq, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                              outs[3].data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for t in s.kernel_times_ms(k):
        seq.append((tag, round(float(t), 3)))


launches("built", 3)
keep = []
for i in range(10):
    t = torch.empty(n * 304 * 4, dtype=torch.uint8, device=dev)
    keep.append(t)
    assert L.cdll.idist_tune_move_buffer(h._h, 0, C.c_void_p(t.data_ptr()), C.c_void_p(orig)) == 0
    launches(f"alloc{i}@{t.data_ptr():x}")
print("SEQ " + json.dumps(seq), flush=True)
