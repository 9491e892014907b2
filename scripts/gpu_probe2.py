"""GPU probe 2 (development aid): build timing vs LDS tile size, CPU-oracle thread scaling."""
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
h = None
for rt in (16, 24, 8):
    os.environ["IDIST_BUILD_RT"] = str(rt)
    t = time.time(); h = ida.Hnsw.from_ordered_points(pts, ida.Builder()); tw = time.time() - t
    st = h.build_stats()
    print(json.dumps({"rt": rt, "build_dev_s": round(st.seconds, 3), "pts_per_s": round(n / st.seconds), "wall": round(tw, 2),
                      "n_heur_dist": st.n_heur_dist, "n_heur_rows": st.n_heur_rows, "n_updates": st.n_updates}), flush=True)
s = ida.Search()
truth, _ = h.bruteforce(q[:500], 10)
r = h.search_batch(q, s, counters=True)
rec = np.mean([len(set(r.pid[i, :10].tolist()) & set(truth[i].tolist())) / 10 for i in range(500)])
print(json.dumps({"recall10": float(rec), "kernel_ms": float(s.kernel_times_ms(1)[0])}), flush=True)

try:
    print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip(), "affinity:", len(os.sched_getaffinity(0)), flush=True)
except Exception as e:  # noqa: BLE001
    print("cgroup:", e)
from oracle import pyoracle as po  # noqa: E402
zero, layers = h.into_parts()
oix = po.Index.from_arrays(pts, zero, layers, po.default_config())
for th in (1, 8, 32, 64, 128, 256):
    nqs = min(10000, 64 * th if th > 1 else 200)
    best = 1e9
    for _ in range(2):
        t = time.time(); oix.search(q[:nqs], threads=th); best = min(best, time.time() - t)
    print(json.dumps({"cpu_threads": th, "nq": nqs, "qps": round(nqs / best, 1)}), flush=True)
