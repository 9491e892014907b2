"""GPU probe 8: full-size CPU build baseline + recall parity of the GPU's batched build at C3.
The threaded oracle (rayon-style, 16 threads = the box's CPU quota) builds the SAME 1M x 300 points; its
graph is imported into the engine and searched with the same queries as the GPU-built graph."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from bench import effective_cores  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim, nq = int(os.environ.get("P8_N", 1_000_000)), 300, 10000
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), nq, dim, "lowrank")
cores = effective_cores()
hg = ida.Hnsw.from_ordered_points(pts, ida.Builder())
st = hg.build_stats()
truth, _ = hg.bruteforce(q, 10)
s = ida.Search()


def recall(h):
    r = h.search_batch(q, s)
    return float(np.mean([len(set(r.pid[i, :10].tolist()) & set(truth[i].tolist())) / 10 for i in range(nq)]))


rg = recall(hg)
zg = hg.into_parts()[0]
print(json.dumps({"gpu_build_s": round(st.seconds, 3), "gpu_pts_per_s": round(n / st.seconds), "gpu_graph_recall10": round(rg, 4),
                  "gpu_graph_mean_degree": float((zg != 0xFFFFFFFF).sum(1).mean())}), flush=True)
t = time.time(); oix = po.Index.build(pts, po.default_config(), threads=cores); tc = time.time() - t
zo, lo = oix.zero, oix.layers
ho = ida.Hnsw.from_parts(pts, zo, lo, ida.Builder())
ro = recall(ho)
print(json.dumps({"cpu_threads": cores, "cpu_build_s": round(tc, 1), "cpu_pts_per_s": round(n / tc), "cpu_graph_recall10": round(ro, 4),
                  "cpu_graph_mean_degree": float((zo != 0xFFFFFFFF).sum(1).mean()),
                  "gpu_over_cpu_build": round((n / st.seconds) / (n / tc), 1)}), flush=True)
