"""GPU probe (development aid): times build + search at growing sizes and prints JSON lines."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402


def gen(rng, n, dim, kind):
    if kind == "uniform":
        return rng.random((n, dim), dtype=np.float32)
    z = rng.standard_normal((n, 32)).astype(np.float32)
    a = np.random.default_rng(4242).standard_normal((32, dim)).astype(np.float32)
    x = z @ a
    x += 0.05 * rng.standard_normal((n, dim)).astype(np.float32) * np.float32(np.sqrt(32))
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def stage(name, n, dim, kind, nq=10000, efs=(100,), gt_q=500, max_batch=0):
    rng = np.random.default_rng(123456789)
    t = time.time(); pts = gen(rng, n, dim, kind); q = gen(np.random.default_rng(123456790), nq, dim, kind)
    t_gen = time.time() - t
    t = time.time(); h = ida.Hnsw.from_ordered_points(pts, ida.Builder().max_batch(max_batch)); t_build = time.time() - t
    st = h.build_stats()
    out = {"stage": name, "n": n, "dim": dim, "kind": kind, "gen_s": round(t_gen, 2), "build_wall_s": round(t_build, 2),
           "build_dev_s": round(st.seconds, 3), "build_pts_per_s": round(n / max(st.seconds, 1e-9)),
           "batches": st.n_batches, "n_dist": st.n_dist, "n_heur_dist": st.n_heur_dist, "n_heur_rows": st.n_heur_rows,
           "n_updates": st.n_updates}
    print(json.dumps(out), flush=True)
    s = ida.Search()
    t = time.time(); truth, _ = h.bruteforce(q[:gt_q], 10); t_bf = time.time() - t
    for ef in efs:
        h.set_ef_search(ef)
        h.search_batch(q[:256], s)
        best = 1e9
        for _ in range(3):
            t = time.time(); r = h.search_batch(q, s, counters=True); best = min(best, time.time() - t)
        from instant_distance_amd import _capi
        import ctypes as C
        ms = C.c_float(0); _capi.lib().check(_capi.lib().idist_search_ctx_last_kernel_ms(s._ctx, C.byref(ms)))
        rec = np.mean([len(set(r.pid[i, :10].tolist()) & set(truth[i].tolist())) / 10 for i in range(gt_q)])
        ctr = r.counters.astype(np.float64).mean(0)
        bq = ctr[0] * 4 * dim + ctr[1] * 256 + ctr[2] * 128 + 8 * ef
        print(json.dumps({"stage": name, "ef": ef, "nq": nq, "wall_qps": round(nq / best), "kernel_ms": round(ms.value, 3),
                          "kernel_qps": round(nq / (ms.value * 1e-3)), "recall10": round(float(rec), 4),
                          "n_dist": round(ctr[0], 1), "n_exp0": round(ctr[1], 1), "n_expU": round(ctr[2], 1),
                          "alg_bytes_per_q": round(bq), "alg_GBps": round(bq * nq / (ms.value * 1e-3) / 1e9, 1),
                          "bf_s": round(t_bf, 2)}), flush=True)
    del h


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "mid", "c3"]
    if "c2" in which:
        stage("c2_100k_128_U", 100_000, 128, "uniform")
    if "c2l" in which:
        stage("c2_100k_128_L", 100_000, 128, "lowrank", efs=(100, 200))
    if "c4" in which:
        stage("c4_1M_768_L", 1_000_000, 768, "lowrank", nq=65536, efs=(100, 200), gt_q=2000)
    if "mid" in which:
        stage("mid_200k_300_L", 200_000, 300, "lowrank", efs=(100, 200))
    if "c3" in which:
        stage("c3_1M_300_L", 1_000_000, 300, "lowrank", efs=(100, 200, 400))
