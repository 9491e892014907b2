"""GPU probe 6: BASELINE config C5's per-GPU part — 10M x 768-d index on ONE MI355X (the 8-GPU run
replicates exactly this index and shards the queries)."""
import json
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from bench import synth  # noqa: E402

n, dim, nq = int(os.environ.get("C5_N", 10_000_000)), 768, 65536
dev = torch.device("cuda", 0)
t = time.time(); d_pts = synth(torch, n, dim, 123456789, dev); torch.cuda.synchronize(); tg = time.time() - t
d_q = synth(torch, nq, dim, 123456790, dev)
t = time.time(); h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder()); tb = time.time() - t
st = h.build_stats()
print(json.dumps({"stage": "c5_10M_768_one_gpu", "gen_s": round(tg, 1), "build_dev_s": round(st.seconds, 2), "pts_per_s": round(n / st.seconds),
                  "wall": round(tb, 1), "fast": st.n_updates_fast, "full": st.n_updates_full, "batches": st.n_batches,
                  "mem_GB": round(torch.cuda.mem_get_info()[0] / 1e9, 1)}), flush=True)
del d_pts
torch.cuda.empty_cache()
q = d_q.cpu().numpy()
t = time.time(); truth, _ = h.bruteforce(q[:1024], 10); tbf = time.time() - t
s = ida.Search()
for ef in (100, 200):
    h.set_ef_search(ef)
    h.search_batch(q[:512], s)
    r = h.search_batch(q, s, counters=True)
    ms = float(s.kernel_times_ms(1)[0])
    rec = np.mean([len(set(r.pid[i, :10].tolist()) & set(truth[i].tolist())) / 10 for i in range(1024)])
    ctr = r.counters.astype(np.float64).mean(0)
    bq = ctr[0] * 4 * dim + ctr[1] * 256 + ctr[2] * 128 + 8 * ef
    print(json.dumps({"stage": "c5_10M_768_one_gpu", "ef": ef, "nq": nq, "kernel_ms": round(ms, 1), "kernel_qps": round(nq / ms * 1e3),
                      "recall10": round(float(rec), 4), "n_dist": round(ctr[0], 1), "n_expU": round(ctr[2], 1),
                      "alg_GBps": round(bq * nq / ms / 1e6, 1), "frac_of_8TBps": round(bq * nq / ms / 1e6 / 8000, 3),
                      "bf_mfma_1024q_s": round(tbf, 2)}), flush=True)
