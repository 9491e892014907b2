"""GPU probe 3: build timing after the memoised update path; row-alignment experiment for search."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim = 1_000_000, 300
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), 10000, dim, "lowrank")
for label, env in (("default", {}), ("align32", {"IDIST_ROW_ALIGN_FLOATS": "32"})):
    for k in ("IDIST_ROW_ALIGN_FLOATS",):
        os.environ.pop(k, None)
    os.environ.update(env)
    t = time.time(); h = ida.Hnsw.from_ordered_points(pts, ida.Builder()); tw = time.time() - t
    st = h.build_stats()
    s = ida.Search()
    truth, _ = h.bruteforce(q[:500], 10)
    for _ in range(3):
        r = h.search_batch(q, s, counters=True)
    rec = np.mean([len(set(r.pid[i, :10].tolist()) & set(truth[i].tolist())) / 10 for i in range(500)])
    print(json.dumps({"cfg": label, "stride": h.info().row_stride, "build_dev_s": round(st.seconds, 3), "pts_per_s": round(n / st.seconds),
                      "fast": st.n_updates_fast, "full": st.n_updates_full, "n_heur_dist": st.n_heur_dist,
                      "recall10": round(float(rec), 4), "search_kernel_ms": [round(float(x), 3) for x in s.kernel_times_ms(3)]}), flush=True)
    del h, s
