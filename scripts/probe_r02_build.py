"""Round-2 GPU probe of the C3 build (1M x 300): device seconds, work counters and recall@10 (1000 held-out queries,
exact ground truth) under the build's schedule knobs.
usage: python scripts/probe_r02_build.py [out.jsonl]   (GPU box)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import instant_distance_amd as ida  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "probe_r02_build.jsonl")
fo = open(out_path, "a")


def emit(**kw):
    print(json.dumps(kw), flush=True)
    fo.write(json.dumps(kw) + "\n")
    fo.flush()


dev = torch.device("cuda", 0)
n, dim = int(os.environ.get("PB_N", 1_000_000)), int(os.environ.get("PB_DIM", 300))
d_pts = bench.synth(torch, n, dim, 123456789, dev)
q = bench.synth(torch, 1000, dim, 123456790, dev).cpu().numpy()
torch.cuda.synchronize()
truth = None
cases = [({}, "default"), ({"IDIST_BUILD_PIPELINE": "0"}, "one stream"),
         ({"IDIST_BUILD_PIPELINE": "0", "IDIST_BUILD_NO_DLOG": "1"}, "one stream, no distance log (step B recomputes)"),
         ({"IDIST_BUILD_NO_DLOG": "1"}, "pipelined, no distance log"), ({"IDIST_BUILD_A_WAVES": "4"}, "4 descent waves per CU"),
         ({"IDIST_BUILD_RT2": "8"}, "A2 tile of 8 selected rows"), ({"IDIST_BUILD_RT2": "4"}, "A2 tile of 4 selected rows"),
         ({"IDIST_BUILD_RT2": "2"}, "A2 tile of 2 selected rows"), ({"IDIST_BUILD_RT2": "4", "IDIST_BUILD_A_WAVES": "4"}, "A2 tile 4, 4 descent waves"),
         ({"IDIST_BUILD_A_WAVES": "2"}, "2 descent waves per CU"),
         ({"IDIST_BUILD_A2_STREAM": "s"}, "step A2 on the update stream (round-1 placement)"),
         ({"IDIST_BUILD_A_WAVES": "4"}, "4 descent waves per CU"), ({"IDIST_BUILD_CHECK": "1"}, "default + pipeline self-check"),
         ({"IDIST_BUILD_A2": "tile"}, "step A2 with the LDS-tile kernel (vector FMA) instead of the Gram matrix on MFMA")]
if os.environ.get("PB_CASES"):
    cases = [cases[int(i)] for i in os.environ["PB_CASES"].split(",")]
for env, nm in cases:
    os.environ.update(env)
    h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
    st = h.build_stats()
    if truth is None:
        truth, _ = h.bruteforce(q, 10)
    got = h.search_batch(q, ida.Search())
    rec = float(np.mean([len(set(got.pid[i, :10].tolist()) & set(truth[i].tolist())) / 10 for i in range(len(q))]))
    emit(case=nm, env=env, seconds=round(st.seconds, 4), points_per_s=round(n / st.seconds), recall_at_10=round(rec, 4),
         n_dist=int(st.n_dist), n_sel_pairs=int(st.n_sel_pairs), n_heur_rows=int(st.n_heur_rows), n_updates=int(st.n_updates),
         n_updates_full=int(st.n_updates_full), batches=int(st.n_batches))
    for k in env:
        os.environ.pop(k)
    del h
