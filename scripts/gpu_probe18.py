"""GPU probe 18: pipelined vs sequential build schedule on C3: build time, recall, determinism."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim, k = 1_000_000, int(os.environ.get("P18_DIM", 300)), 10
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), 10000, dim, "lowrank")
truth = None
prev = {}
for rnd in range(2):
    for pipe, aw in (("1", "16"), ("0", "16")):
        os.environ["IDIST_BUILD_PIPELINE"] = pipe
        h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
        st = h.build_stats()
        if truth is None:
            truth, _ = h.bruteforce(q, k)
        s = ida.Search()
        for _ in range(3):
            r = h.search_batch(q, s, counters=True)
        rec = float(np.mean([len(set(r.pid[i, :k].tolist()) & set(truth[i].tolist())) / k for i in range(len(q))]))
        zero, _ = h.into_parts()
        deg = float((zero != 0xFFFFFFFF).sum(axis=1).mean())
        same = None if pipe not in prev else bool(np.array_equal(prev[pipe], zero))
        prev[pipe] = zero
        rec = rec if aw in ("16",) else -1.0
        print(json.dumps({"pipeline": pipe, "a_waves": aw, "round": rnd, "build_s": round(st.seconds, 4), "recall10_ef100": round(rec, 4), "mean_degree": round(deg, 2),
                          "search_ms": round(float(np.median(s.kernel_times_ms(2))), 3), "n_dist_per_q": float(r.counters[:, 0].mean()),
                          "same_graph_as_previous_round": same, "n_updates_full": int(st.n_updates_full)}), flush=True)
        del h, s
