"""GPU probe 11: latency / throughput of the search path as a function of the batch width (C3 index).
nq = 1 is the reference's own call pattern (`Hnsw::search`, one query per call)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim = 1_000_000, int(os.environ.get("P11_DIM", 300))
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), 16384, dim, "lowrank")
h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
print(json.dumps({"build_s": round(h.build_stats().seconds, 3)}), flush=True)
s = ida.Search()
for ef in (100, 400):
    h.set_ef_search(ef)
    for nq in (1, 2, 8, 32, 128, 512, 2048, 4096, 8192, 16384):
        reps = 40 if nq <= 128 else 8
        walls = []
        for i in range(reps):
            qq = q[(i * nq) % (16384 - nq + 1):][:nq]
            t0 = time.perf_counter()
            h.search_batch(qq, s)
            walls.append(time.perf_counter() - t0)
        ms = s.kernel_times_ms(min(reps - 2, 32))
        print(json.dumps({"ef": ef, "nq": nq, "kernel_ms_med": round(float(np.median(ms)), 4), "kernel_ms_min": round(float(ms.min()), 4),
                          "wall_ms_med": round(float(np.median(walls[2:])) * 1e3, 4),
                          "kernel_qps": round(nq / (float(np.median(ms)) * 1e-3))}), flush=True)
