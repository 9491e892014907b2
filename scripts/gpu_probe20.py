"""GPU probe 20: does the order of the queries inside a batch matter?  Consecutive queries run concurrently (the
work queue hands them out in index order), so sorting the batch by a coarse cell makes concurrent walks share
rows in L2 / Infinity Cache.  Results are per-query, hence unchanged."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim = 1_000_000, 300
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), 16384, dim, "lowrank")
h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
s = ida.Search()


def run(qq, tag):
    for _ in range(6):
        r = h.search_batch(qq, s)
    ms = s.kernel_times_ms(5)
    print(json.dumps({"order": tag, "nq": len(qq), "kernel_ms_min": round(float(ms.min()), 3), "kernel_ms_med": round(float(np.median(ms)), 3)}), flush=True)
    return r


for rnd in range(2):
    r0 = run(q, "as given")
    for cells in (256, 2048, 16384):
        cent = pts[:cells]                                   # the first pids = the upper layers' points
        d = (q * q).sum(1)[:, None] - 2.0 * q @ cent.T + (cent * cent).sum(1)[None, :]
        cell = d.argmin(1)
        order = np.argsort(cell, kind="stable")
        r = run(np.ascontiguousarray(q[order]), f"sorted by nearest of the first {cells} points")
        assert np.array_equal(r.pid, r0.pid[order])
    # the limit: sort by the true nearest neighbour's id is meaningless (ids are random); sort by first result of a coarse search
    order = np.lexsort((r0.pid[:, 1], r0.pid[:, 0]))
    run(np.ascontiguousarray(q[order]), "sorted by own nearest neighbour id (oracle order, upper bound of cell sorting)")
