"""GPU probe 9: A/B of the LDS Bloom filter in front of the visited bytes (IDIST_BLOOM=0/1), search + build."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim = 1_000_000, int(os.environ.get("P9_DIM", 300))
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), 10000, dim, "lowrank")
for rnd in range(2):
    for bloom in ("1", "0"):
        os.environ["IDIST_BLOOM"] = bloom
        h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
        st = h.build_stats()
        s = ida.Search()
        for _ in range(6):
            r = h.search_batch(q, s, counters=True)
        ms = s.kernel_times_ms(5)
        print(json.dumps({"bloom": bloom, "round": rnd, "dim": dim, "build_s": round(st.seconds, 3), "search_ms_min": round(float(ms.min()), 3),
                          "search_ms_med": round(float(np.median(ms)), 3), "n_dist": float(r.counters[:, 0].mean())}), flush=True)
        del h, s
