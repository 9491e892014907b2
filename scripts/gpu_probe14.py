"""GPU probe 14: build step cap (Builder.max_batch): build time vs graph quality on C3."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim, k = 1_000_000, 300, 10
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), 10000, dim, "lowrank")
truth = None
for cap in (8192, 4096, 16384, 32768, 8192):
    h = ida.Hnsw.from_ordered_points(pts, ida.Builder().max_batch(cap))
    st = h.build_stats()
    if truth is None:
        truth, _ = h.bruteforce(q, k)
    s = ida.Search()
    for _ in range(4):
        r = h.search_batch(q, s, counters=True)
    rec = float(np.mean([len(set(r.pid[i, :k].tolist()) & set(truth[i].tolist())) / k for i in range(len(q))]))
    print(json.dumps({"max_batch": cap, "build_s": round(st.seconds, 3), "batches": int(st.n_batches), "recall10_ef100": round(rec, 4),
                      "search_ms": round(float(np.median(s.kernel_times_ms(3))), 3), "n_dist_per_q": float(r.counters[:, 0].mean()),
                      "n_updates_full": int(st.n_updates_full)}), flush=True)
    del h, s
