"""Quality of the DEFAULT (batched) GPU build against the CPU oracle's threaded build (per-layer parallel-for + per-node locks = the
reference's rayon path, core/lib.rs:316-318) on the same points: both graphs are searched BY THE ENGINE with the same queries, so
the only difference is who built the graph.  The reference's concurrent build is non-deterministic; recall and degree statistics
are the only yardstick the batched schedule can be held to (tier 3).
usage: python scripts/probe_build_quality.py out.jsonl   (PB_N / PB_DIM / PB_NQ / PB_THREADS; default C3: 1M x 300, 10k queries)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import instant_distance_amd as ida  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

fo = open(sys.argv[1], "a")
dev = torch.device("cuda", 0)
n, dim, nq = int(os.environ.get("PB_N", 1_000_000)), int(os.environ.get("PB_DIM", 300)), int(os.environ.get("PB_NQ", 10_000))
threads = int(os.environ.get("PB_THREADS", 0)) or bench.effective_cores()
d_pts = bench.synth(torch, n, dim, 123456789, dev)
q = bench.synth(torch, nq, dim, 123456790, dev).cpu().numpy()
torch.cuda.synchronize()
pts = d_pts.cpu().numpy()


def describe(who, h, seconds, truth):
    zero, layers = h.into_parts()
    deg = (zero != 0xFFFFFFFF).sum(1)
    row = dict(probe="build_quality", commit=bench.source_stamp(), builder=who, n=n, dim=dim, queries=nq, build_seconds=round(seconds, 3),
               mean_degree_zero=round(float(deg.mean()), 3), min_degree_zero=int(deg.min()), rows_full_zero=int((deg == 64).sum()),
               mean_degree_upper=[round(float((l != 0xFFFFFFFF).sum(1).mean()), 3) for l in layers])
    for ef in (100, 200):
        h.set_ef_search(ef)
        got = h.search_batch(q, ida.Search(), counters=True)
        row[f"recall_at_10_ef{ef}"] = round(float(np.mean([len(set(got.pid[i, :10].tolist()) & set(truth[i].tolist())) / 10 for i in range(nq)])), 4)
        row[f"n_dist_per_query_ef{ef}"] = round(float(got.counters[:, 0].mean()), 1)
    print(json.dumps(row), flush=True)
    fo.write(json.dumps(row) + "\n")
    fo.flush()


g = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
truth, _ = g.bruteforce(q, 10)
describe("gpu default (batched) schedule", g, g.build_stats().seconds, truth)
del g
t0 = time.time()
oix = po.Index.build(pts, po.default_config(), threads=threads)
t_cpu = time.time() - t0
c = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, ida.Builder())
describe(f"cpu oracle, {threads} threads (restated reference, rayon-style)", c, t_cpu, truth)
