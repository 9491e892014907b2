"""GPU probe 7: search kernel time vs resident query slots (occupancy sensitivity)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim = 1_000_000, 300
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), 10000, dim, "lowrank")
h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
for slots in (512, 1024, 1536, 2048, 3072, 4096, 6144, 8192):
    s = ida.Search(slots=slots)
    for _ in range(4):
        h.search_batch(q, s)
    ms = s.kernel_times_ms(3)
    print(json.dumps({"slots": slots, "waves_per_cu": slots / 256, "kernel_ms": [round(float(x), 2) for x in ms]}), flush=True)
    del s
