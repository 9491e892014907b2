"""Search kernel ms (10k-query launches, HIP events, best of 3) under knob cases, one process, one index: PB_N / PB_DIM / PB_EFS;
cases as "name:KEY=VAL,KEY=VAL" ("default:" = no knob).  Loads the TEST build (the knobs exist there only).
usage: python scripts/probe_search_knobs.py out.jsonl case [case ...]"""
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
_capi._singleton = _capi.Lib(os.path.join(os.path.dirname(_capi.LIB_PATH), os.environ.get("PB_LIB", "libidist_variants.so")))
fo = open(sys.argv[1], "a")
dev = torch.device("cuda", 0)
n, dim, nq = int(os.environ.get("PB_N", 1_000_000)), int(os.environ.get("PB_DIM", 300)), 10_000
d_pts = bench.synth(torch, n, dim, 123456789, dev)
d_q = bench.synth(torch, nq, dim, 123456790, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
job = bench.Job(torch, dev=dev)
for ef in [int(x) for x in os.environ.get("PB_EFS", "600,800").split(",")]:
    h.set_ef_search(ef)
    row = dict(probe="search_knobs", commit=bench.source_stamp(), n=n, dim=dim, ef=ef)
    for spec in sys.argv[2:]:
        nm, _, kv = spec.partition(":")
        env = dict(x.split("=", 1) for x in kv.split(",") if x)
        os.environ.update(env)
        r = bench.Runner(job, ida, h, d_q)          # a fresh context samples the knobs
        outs = r.alloc_out(ef)
        r.run(outs)
        torch.cuda.synchronize()
        r.search.filter_counts()                    # what the reject filter examined / rejected: reset
        for _ in range(3):
            r.run(outs)
        torch.cuda.synchronize()
        r.search.check_status()
        seen, rej = r.search.filter_counts()
        row[nm + "_filter_rejected_share"] = round(rej / seen, 4) if seen else None
        row[nm + "_ms"] = round(float(r.search.kernel_times_ms(3).min()), 3)
        row[nm + "_checksum"] = int(outs[0].to(torch.int64).sum().item())
        for k in env:
            os.environ.pop(k, None)
        del r, outs
    print(json.dumps(row), flush=True)
    fo.write(json.dumps(row) + "\n")
    fo.flush()
