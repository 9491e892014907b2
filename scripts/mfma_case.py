"""C4 distance path under the profiler: exact 10-NN of 4096 queries against 1M x 768 through the
f32-MFMA -2QP^T filter (mfma_dist_kernel) + canonical re-rank."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim, nq = 1_000_000, 768, 4096
pts = gen(np.random.default_rng(1), n, dim, "lowrank")
q = gen(np.random.default_rng(2), nq, dim, "lowrank")
h = ida.Hnsw.from_parts(pts, np.full((n, 64), 0xFFFFFFFF, np.uint32), [], ida.Builder())
os.environ["IDIST_BRUTEFORCE"] = "mfma"
for _ in range(3):
    t = time.time(); pid, d = h.bruteforce(q, 10); dt = time.time() - t
print("mfma bruteforce", nq, "x", n, "x", dim, "wall_s", round(dt, 4), "flop", 2.0 * nq * n * h.info().row_stride * (1 + 32768 / n))
