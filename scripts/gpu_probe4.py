"""GPU probe 4: build after cascade memo; MFMA brute force throughput; C4 (1M x 768, 64k queries)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402


def stage(name, n, dim, nq, efs, gtq):
    pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
    q = gen(np.random.default_rng(123456790), nq, dim, "lowrank")
    t = time.time(); h = ida.Hnsw.from_ordered_points(pts, ida.Builder()); tw = time.time() - t
    st = h.build_stats()
    print(json.dumps({"stage": name, "build_dev_s": round(st.seconds, 3), "pts_per_s": round(n / st.seconds), "wall": round(tw, 2),
                      "fast": st.n_updates_fast, "full": st.n_updates_full, "n_heur_dist": st.n_heur_dist}), flush=True)
    os.environ["IDIST_BRUTEFORCE"] = "mfma"
    t = time.time(); truth, td = h.bruteforce(q[:gtq], 10); t_m = time.time() - t
    os.environ["IDIST_BRUTEFORCE"] = "scan"
    t = time.time(); truth_s, _ = h.bruteforce(q[:512], 10); t_s = time.time() - t
    print(json.dumps({"stage": name, "bf_mfma_s": round(t_m, 3), "bf_mfma_queries": gtq,
                      "bf_mfma_TFLOPs_incl_host": round(2.0 * gtq * n * h.info().row_stride / t_m / 1e12, 2),
                      "bf_scan_s_512q": round(t_s, 3), "mfma_equals_scan": bool(np.array_equal(truth[:512], truth_s))}), flush=True)
    s = ida.Search()
    for ef in efs:
        h.set_ef_search(ef)
        h.search_batch(q[:256], s)
        for _ in range(2):
            t = time.time(); r = h.search_batch(q, s, counters=True); tq = time.time() - t
        ms = float(s.kernel_times_ms(1)[0])
        rec = np.mean([len(set(r.pid[i, :10].tolist()) & set(truth[i].tolist())) / 10 for i in range(gtq)])
        ctr = r.counters.astype(np.float64).mean(0)
        bq = ctr[0] * 4 * dim + ctr[1] * 256 + ctr[2] * 128 + 8 * ef
        print(json.dumps({"stage": name, "ef": ef, "nq": nq, "kernel_ms": round(ms, 2), "kernel_qps": round(nq / ms * 1e3),
                          "wall_qps": round(nq / tq), "recall10": round(float(rec), 4), "n_dist": round(ctr[0], 1),
                          "alg_GBps": round(bq * nq / ms / 1e6, 1), "frac_of_8TBps": round(bq * nq / ms / 1e6 / 8000, 3)}), flush=True)


which = sys.argv[1:] or ["c3", "c4"]
if "c3" in which:
    stage("c3_1M_300", 1_000_000, 300, 10000, (100,), 4096)
if "c4" in which:
    stage("c4_1M_768", 1_000_000, 768, 65536, (100, 200), 4096)
