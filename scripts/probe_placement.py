"""Long walks (ef_search 800 at C3) differ by ~10 % between fresh processes on one box.  Which allocation carries the effect?  One
process, one built index: K fresh search contexts (each allocates its own visited bitmaps) on the same index, then K replicas of the
index on the same device (each allocates points / zero / upper anew) searched through fresh contexts — kernel ms of 10k-query
launches (HIP events), the device addresses of the index buffers, identical answers throughout.
usage: python scripts/probe_placement.py out.jsonl   (PB_EF, default 800; PB_K, default 5)"""
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

fo = open(sys.argv[1], "a")
dev = torch.device("cuda", 0)
n, dim, nq = int(os.environ.get("PB_N", 1_000_000)), int(os.environ.get("PB_DIM", 300)), 10_000
ef, K = int(os.environ.get("PB_EF", 800)), int(os.environ.get("PB_K", 5))
d_pts = bench.synth(torch, n, dim, 123456789, dev)
d_q = bench.synth(torch, nq, dim, 123456790, dev)
torch.cuda.synchronize()
root = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
job = bench.Job(torch, dev=dev)


def addresses(h):
    b = _capi.DeviceBuffers()
    _capi.lib().check(_capi.lib().idist_index_device_buffers(h._h, C.byref(b)))
    return {"points": hex(b.points), "zero": hex(b.zero), "upper": hex(b.upper)}


def measure(h, what, i, keep):
    h.set_ef_search(ef)
    r = bench.Runner(job, ida, h, d_q)
    outs = r.alloc_out(ef)
    for _ in range(4):
        r.run(outs)
    torch.cuda.synchronize()
    r.search.check_status()
    kt = r.search.kernel_times_ms(3)
    ctr = outs[3].cpu().numpy().astype(np.int64)
    nbytes = int((ctr[:, 0] * 4 * dim + ctr[:, 1] * 256 + ctr[:, 2] * 128 + 8 * ef).sum())
    row = dict(probe="placement", commit=bench.source_stamp(), what=what, i=i, ef=ef, kernel_ms=[round(float(x), 3) for x in kt],
               frac_of_8TBps=round(nbytes / (float(kt.min()) * 1e-3) / 8e12, 4), index_buffers=addresses(h),
               answers_checksum=int(outs[0].to(torch.int64).sum().item()))
    print(json.dumps(row), flush=True)
    fo.write(json.dumps(row) + "\n")
    fo.flush()
    keep.append((r, outs))          # contexts and buffers stay allocated: the next one lands somewhere else


keep = []
for i in range(K):
    measure(root, "fresh context, same index", i, keep)
reps = []
for i in range(K):
    reps.append(root.replicate([0])[0])
    measure(reps[-1], "fresh replica of the index on the same device, fresh context", i, keep)
measure(root, "the root index again, fresh context", K, keep)
