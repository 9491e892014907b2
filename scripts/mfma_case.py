"""C4 distance path under the profiler: exact 10-NN of 4096 queries against 1M x 768 through the
f32-MFMA -2QP^T filter (mfma_dist_kernel) + canonical re-rank.  usage: python scripts/mfma_case.py   (GPU box)"""
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

if os.environ.get("PB_LIB"):          # an alternative build of the library (A/B of a kernel change in one box session)
    _capi._singleton = _capi.Lib(os.path.join(ROOT, "instant-distance_amd", "csrc", os.environ["PB_LIB"]))

dev = torch.device("cuda", 0)
n, dim, nq = 1_000_000, 768, 4096
pts = bench.synth(torch, n, dim, 1, dev).cpu().numpy()
q = bench.synth(torch, nq, dim, 2, dev).cpu().numpy()
h = ida.Hnsw.from_parts(pts, np.full((n, 64), 0xFFFFFFFF, np.uint32), [], ida.Builder())
os.environ["IDIST_BRUTEFORCE"] = "mfma"
for _ in range(3):
    t = time.time(); pid, d = h.bruteforce(q, 10); dt = time.time() - t
stride = h.info().row_stride
print(json.dumps({"case": "mfma bruteforce", "nq": nq, "n": n, "dim": dim, "wall_s": round(dt, 4),
                  "flop_filter_pass": 2.0 * nq * n * stride, "flop_sample_pass": 2.0 * nq * 32768 * stride}))
