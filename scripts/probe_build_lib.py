"""C3 build seconds with an alternative library (tuning builds).  usage: python scripts/probe_build_lib.py <lib.so> [env=val ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    os.environ[k] = v
import torch  # noqa: E402

import bench  # noqa: E402
import instant_distance_amd as ida  # noqa: E402
from instant_distance_amd import _capi  # noqa: E402

_capi._singleton = _capi.Lib(os.path.join(ROOT, "instant-distance_amd", "csrc", sys.argv[1]))
dev = torch.device("cuda", 0)
d_pts = bench.synth(torch, 1_000_000, 300, 123456789, dev)
torch.cuda.synchronize()
for _ in range(2):
    h = ida.Hnsw.from_device_points(d_pts.data_ptr(), 1_000_000, 300, ida.Builder())
    print(json.dumps({"lib": sys.argv[1], "env": sys.argv[2:], "build_s": round(h.build_stats().seconds, 4)}), flush=True)
    del h
