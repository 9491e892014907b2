"""One C3 build (for rocprofv3 kernel traces of the build schedule).  usage: python scripts/build_only.py   (GPU box)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import instant_distance_amd as ida  # noqa: E402

dev = torch.device("cuda", 0)
n, dim = int(os.environ.get("PB_N", 1_000_000)), int(os.environ.get("PB_DIM", 300))
d_pts = bench.synth(torch, n, dim, 123456789, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
print(json.dumps({"build_s": round(h.build_stats().seconds, 4), "pipeline": os.environ.get("IDIST_BUILD_PIPELINE", "1")}))
