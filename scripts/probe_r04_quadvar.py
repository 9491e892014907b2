"""One query per call and a 100k x 128-d build with one build of the library (a protocol variant of the four-wave walk,
-DIDIST_QV=n: bit 0 adjacency two expansions ahead, bit 1 next candidate known before the merge, bit 2 a new candidate's row
handed over after the push, bit 3 helpers never work ahead).  One process per library, same box, same index and queries.
usage: python scripts/probe_r04_quadvar.py <lib.so> [out.jsonl]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import instant_distance_amd as ida  # noqa: E402
from instant_distance_amd import _capi  # noqa: E402

lib = sys.argv[1]
_capi._singleton = _capi.Lib(lib)
out = open(sys.argv[2], "a") if len(sys.argv) > 2 else sys.stdout
dev = torch.device("cuda", 0)
rec = {"probe": "quad_walk_variant", "commit": bench.source_stamp(), "lib": os.path.basename(lib)}
calls = 256
for dim, n in ((300, 1_000_000), (768, 200_000)):
    d_pts = bench.synth(torch, n, dim, 123456789, dev)
    d_q = bench.synth(torch, calls, dim, 123456790, dev)
    torch.cuda.synchronize()
    h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
    rec[f"build_{n}x{dim}_s"] = round(h.build_stats().seconds, 4)
    for ef in (100, 400):
        h.set_ef_search(ef)
        s = ida.Search()
        o = (torch.empty(1, ef, dtype=torch.int32, device=dev), torch.empty(1, ef, dtype=torch.float32, device=dev),
             torch.empty(1, dtype=torch.int32, device=dev), torch.empty(1, 3, dtype=torch.int32, device=dev))
        stream = torch.cuda.current_stream().cuda_stream
        times, digest = [], 0
        for i in range(calls + 8):
            h.search_batch_device(s, d_q.data_ptr() + 4 * dim * (i % calls), 1, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), stream)
            torch.cuda.synchronize()
            if i >= 8:
                times.append(float(s.kernel_times_ms(1)[-1]))
                digest = (digest * 1000003 + int(o[0].to(torch.int64).sum().item()) + int(o[3].to(torch.int64).sum().item())) % (1 << 61)
        s.check_status()
        rec[f"one_query_ms_{dim}d_ef{ef}"] = round(float(np.median(times)), 4)
        rec[f"digest_{dim}d_ef{ef}"] = digest
    del h, d_pts
    torch.cuda.empty_cache()
d_pts = bench.synth(torch, 100_000, 128, 5, dev)
best = 1e9
for _ in range(3):
    h = ida.Hnsw.from_device_points(d_pts.data_ptr(), 100_000, 128, ida.Builder())
    best = min(best, h.build_stats().seconds)
rec["build_100000x128_s"] = round(best, 4)
print(json.dumps(rec), file=out, flush=True)
