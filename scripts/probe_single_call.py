"""Wall time of the reference's call pattern — one host query per Hnsw::search through idist_search_batch with host pointers —
with and without the zero-copy path (IDIST_NO_ZERO_COPY), C3.  usage: python scripts/probe_single_call.py   (GPU box)"""
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

# the knobs this script steers with exist in the test build only (libidist_variants.so); PB_LIB names another library
torch.cuda.init()
_capi._singleton = _capi.Lib(os.path.join(ROOT, "instant-distance_amd", "csrc", os.environ.get("PB_LIB", "libidist_variants.so")))

dev = torch.device("cuda", 0)
n, dim = 1_000_000, 300
d_pts = bench.synth(torch, n, dim, 123456789, dev)
q = bench.synth(torch, 1024, dim, 123456790, dev).cpu().numpy()
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
ref = {}
for nm, env in (("staged copies", {"IDIST_NO_ZERO_COPY": "1"}), ("zero-copy", {}), ("staged copies", {"IDIST_NO_ZERO_COPY": "1"}), ("zero-copy", {})):
    os.environ.pop("IDIST_NO_ZERO_COPY", None)
    os.environ.update(env)
    s = ida.Search()
    h.search_batch(q[:1], s)
    for width in (1, 8, 64, 256, 512):
        t0 = time.perf_counter()
        res = [h.search_batch(q[i:i + width], s, counters=True) for i in range(0, 512, width)]
        wall = (time.perf_counter() - t0) / (512 / width) * 1e3
        kt = s.kernel_times_ms(64)
        pid = np.concatenate([r.pid for r in res])
        ref.setdefault(width, pid)
        print(json.dumps({"path": nm, "queries_per_call": width, "wall_ms_per_call": round(wall, 4),
                          "kernel_ms_median": round(float(np.median(kt)), 4), "same_results": bool(np.array_equal(pid, ref[width]))}), flush=True)
    del s
os.environ.pop("IDIST_NO_ZERO_COPY", None)
