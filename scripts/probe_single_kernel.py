"""One query per launch (the reference's Hnsw::search) and one wide batch, any library, any dimension: median kernel ms of 64 distinct
single-query launches, kernel ms of a 10k-query launch, a checksum of the answers.  PB_LIB / PB_N / PB_DIM / PB_EF.
usage: python scripts/probe_single_kernel.py out.jsonl"""
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

torch.cuda.init()
lib = os.environ.get("PB_LIB", "libidist.so")
_capi._singleton = _capi.Lib(os.path.join(ROOT, "instant-distance_amd", "csrc", lib))
fo = open(sys.argv[1], "a")
dev = torch.device("cuda", 0)
n, dim, nq, ef = int(os.environ.get("PB_N", 1_000_000)), int(os.environ.get("PB_DIM", 300)), 10_000, int(os.environ.get("PB_EF", 100))
d_pts = bench.synth(torch, n, dim, 123456789, dev)
d_q = bench.synth(torch, nq, dim, 123456790, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
h.set_ef_search(ef)
job = bench.Job(torch, dev=dev)
r = bench.Runner(job, ida, h, d_q)
outs = r.alloc_out(ef)
for _ in range(4):
    r.run(outs)
torch.cuda.synchronize()
wide = float(r.search.kernel_times_ms(3).min())
chk = int(outs[0].to(torch.int64).sum().item())
for i in range(64):
    r.run(outs, d_q=d_q[i:i + 1])
torch.cuda.synchronize()
r.search.check_status()
single = r.search.kernel_times_ms(64)
row = dict(probe="single_kernel", commit=bench.source_stamp(), lib=lib, n=n, dim=dim, ef=ef, build_s=round(h.build_stats().seconds, 4),
           wide_10k_kernel_ms=round(wide, 3), single_query_kernel_ms_median=round(float(np.median(single)), 4),
           single_query_kernel_ms_min=round(float(single.min()), 4), answers_checksum=chk)
print(json.dumps(row), flush=True)
fo.write(json.dumps(row) + "\n")
