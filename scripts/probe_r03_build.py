"""Round-3 GPU probe of the build (C3 1M x 300 by default; PB_N / PB_DIM for C4's 768-d): device seconds, work counters and
recall@10 under the descent's visited-set form (16-KB quotient set vs 32-KB id set), its register budget (one fat wave per
SIMD vs two 256-register waves) and the descent waves per CU beside the update stream.
usage: python scripts/probe_r03_build.py out.jsonl [case,case,...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import instant_distance_amd as ida  # noqa: E402

fo = open(sys.argv[1], "a")
dev = torch.device("cuda", 0)
n, dim = int(os.environ.get("PB_N", 1_000_000)), int(os.environ.get("PB_DIM", 300))
d_pts = bench.synth(torch, n, dim, 123456789, dev)
q = bench.synth(torch, 1000, dim, 123456790, dev).cpu().numpy()
torch.cuda.synchronize()
CASES = {
    "default": {},
    "ids3": {"IDIST_TAB_FORMAT": "ids"},                                       # round 2: 32-KB id set, three descent waves per CU
    "q16w3": {"IDIST_BUILD_A_WAVES": "3"},
    "r256w4": {"IDIST_BUILD_A_REGS": "256", "IDIST_BUILD_A_WAVES": "4"},
    "r256w5": {"IDIST_BUILD_A_REGS": "256", "IDIST_BUILD_A_WAVES": "5"},
    "r256w6": {"IDIST_BUILD_A_REGS": "256", "IDIST_BUILD_A_WAVES": "6"},
    "r256w7": {"IDIST_BUILD_A_REGS": "256", "IDIST_BUILD_A_WAVES": "7"},
    "onestream": {"IDIST_BUILD_PIPELINE": "0"},
    "onestream_r256": {"IDIST_BUILD_PIPELINE": "0", "IDIST_BUILD_A_REGS": "256"},
    "a2tile": {"IDIST_BUILD_A2": "tile"},
    "noquad": {"IDIST_BUILD_QUAD": "0"},
    "seq20k": {"PB_MAX_BATCH": "1", "PB_N_SEQ": "20000"},
    "seq20k_noquad": {"PB_MAX_BATCH": "1", "PB_N_SEQ": "20000", "IDIST_BUILD_QUAD": "0"},
    "chunk4": {"IDIST_BUILD_CHUNK": "4"},
    "default2": {},
}
names = sys.argv[2].split(",") if len(sys.argv) > 2 else list(CASES)
truth = None
for nm in names:
    env = CASES[nm]
    os.environ.update(env)
    try:
        nn = int(env.get("PB_N_SEQ", n))                       # sequential (max_batch = 1) cases build a prefix
        h = ida.Hnsw.from_device_points(d_pts.data_ptr(), nn, dim, ida.Builder().max_batch(int(env.get("PB_MAX_BATCH", 0))))
        st = h.build_stats()
        if truth is None:
            truth, _ = h.bruteforce(q, 10)
        got = h.search_batch(q, ida.Search())
        rec = float(np.mean([len(set(got.pid[i, :10].tolist()) & set(truth[i].tolist())) / 10 for i in range(len(q))]))
        zero, _ = h.into_parts()
        row = dict(case=nm, n=nn, dim=dim, env=env, seconds=round(st.seconds, 4), points_per_s=round(nn / st.seconds), recall_at_10=round(rec, 4),
                   n_dist=int(st.n_dist), n_sel_pairs=int(st.n_sel_pairs), n_heur_rows=int(st.n_heur_rows), n_updates=int(st.n_updates),
                   n_updates_full=int(st.n_updates_full), batches=int(st.n_batches), graph_checksum=int(zero.astype(np.uint64).sum()))
        del h, zero
    except Exception as e:  # noqa: BLE001
        row = dict(case=nm, env=env, err=repr(e)[:200])
    for k in env:
        os.environ.pop(k)
    print(json.dumps(row), flush=True)
    fo.write(json.dumps(row) + "\n")
    fo.flush()
