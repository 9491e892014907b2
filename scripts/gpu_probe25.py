"""GPU probe 25: placement effect, part 2: ONE index searched through 5 Search contexts (own visited arrays / scratch
each), then 4 imports of the same graph searched through fresh contexts — which allocation carries the effect?"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim, nq = 1_000_000, 300, 10000
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), nq, dim, "lowrank")
h0 = ida.Hnsw.from_ordered_points(pts, ida.Builder())
zero, layers = h0.into_parts()
ctxs = [ida.Search() for _ in range(5)]
times = [[] for _ in ctxs]
for rep in range(10):
    for i, s in enumerate(ctxs):
        h0.search_batch(q, s)
        times[i].append(float(s.kernel_times_ms(1)[0]))
print(json.dumps({"one index, five contexts: ms_median": [round(float(np.median(t[2:])), 3) for t in times]}), flush=True)
hs = [ida.Hnsw.from_parts(pts, zero, layers, ida.Builder()) for _ in range(4)]
for rnd in range(2):
    ss = [ida.Search() for _ in hs]
    times = [[] for _ in hs]
    for rep in range(10):
        for i, (h, s) in enumerate(zip(hs, ss)):
            h.search_batch(q, s)
            times[i].append(float(s.kernel_times_ms(1)[0]))
    print(json.dumps({"four imports, fresh contexts (round %d): ms_median" % rnd: [round(float(np.median(t[2:])), 3) for t in times]}), flush=True)
    del ss
