"""Build the C3 index once (for kernel traces of the build schedule)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim = int(os.environ.get("B1_N", 1_000_000)), int(os.environ.get("B1_DIM", 300))
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
print(json.dumps({"build_s": round(h.build_stats().seconds, 4), "pipeline": os.environ.get("IDIST_BUILD_PIPELINE", "1")}))
