"""One fresh process: C3 index (PB_N / PB_DIM), 10k-query launches at each ef_search of PB_EFS (default 400,800): kernel ms (HIP events,
best of 3 after the first), fraction of 8 TB/s, a checksum of the answers.  Run it several times from a shell loop to see the spread
between fresh processes.  usage: python scripts/probe_ef_once.py out.jsonl"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import instant_distance_amd as ida  # noqa: E402
from instant_distance_amd import _capi  # noqa: E402

if os.environ.get("PB_LIB"):          # an alternative build of the library (A/B in one box session)
    torch.cuda.init()
    _capi._singleton = _capi.Lib(os.path.join(ROOT, "instant-distance_amd", "csrc", os.environ["PB_LIB"]))

fo = open(sys.argv[1], "a")
dev = torch.device("cuda", 0)
n, dim, nq = int(os.environ.get("PB_N", 1_000_000)), int(os.environ.get("PB_DIM", 300)), 10_000
d_pts = bench.synth(torch, n, dim, 123456789, dev)
d_q = bench.synth(torch, nq, dim, 123456790, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
job = bench.Job(torch, dev=dev)
for ef in [int(x) for x in os.environ.get("PB_EFS", "400,800").split(",")]:
    h.set_ef_search(ef)
    r = bench.Runner(job, ida, h, d_q)
    outs = r.alloc_out(ef)
    t0 = time.perf_counter()
    r.run(outs)
    torch.cuda.synchronize()
    first_wall = time.perf_counter() - t0
    for _ in range(3):
        r.run(outs)
    torch.cuda.synchronize()
    r.search.check_status()
    kt = r.search.kernel_times_ms(3)
    ctr = outs[3].cpu().numpy().astype(np.int64)
    nbytes = int((ctr[:, 0] * 4 * dim + ctr[:, 1] * 256 + ctr[:, 2] * 128 + 8 * ef).sum())
    row = dict(probe="ef_once", commit=bench.source_stamp(), lib=os.environ.get("PB_LIB", "libidist.so"), build_s=round(h.build_stats().seconds, 4), pid=os.getpid(), n=n, dim=dim, ef=ef, kernel_ms=[round(float(x), 3) for x in kt],
               frac_of_8TBps=round(nbytes / (float(kt.min()) * 1e-3) / 8e12, 4), first_call_wall_ms=round(first_wall * 1e3, 1),
               answers_checksum=int(outs[0].to(torch.int64).sum().item()))
    print(json.dumps(row), flush=True)
    fo.write(json.dumps(row) + "\n")
    fo.flush()
