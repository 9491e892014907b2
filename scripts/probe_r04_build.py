"""Round-4 GPU probe of the build schedule (C3 1M x 300 by default; PB_N / PB_DIM): device seconds, work counters, recall@10
and a checksum of the graph under the stream layout (descents of odd / even steps on their own streams; the new points'
selection on its own stream beside the previous step's updates) and the step cap.  One process, one data set, every case
PB_REPS times.
usage: python scripts/probe_r04_build.py out.jsonl [case,case,...]"""
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
reps = int(os.environ.get("PB_REPS", 3))
d_pts = bench.synth(torch, n, dim, 123456789, dev)
q = bench.synth(torch, 2000, dim, 123456790, dev).cpu().numpy()
torch.cuda.synchronize()
CASES = {
    "default": {},                                                              # extra streams in narrow steps only
    "streams_off": {"IDIST_BUILD_STREAMS": "off"},                             # round 3's layout: one descent stream, one update stream
    "streams_all": {"IDIST_BUILD_STREAMS": "all"},                             # extra streams in every step
    "cap16384": {"PB_MAX_BATCH": "16384"},
    "cap32768": {"PB_MAX_BATCH": "32768"},
    "a_waves5": {"IDIST_BUILD_A_WAVES": "5"},
    "a_waves3": {"IDIST_BUILD_A_WAVES": "3"},
    "check": {"IDIST_BUILD_CHECK": "1"},                                        # both zero-layer copies must agree at the end
    "no_dlog": {"IDIST_BUILD_NO_DLOG": "1"},
    "no_quad": {"IDIST_BUILD_QUAD": "0"},
    "growth16": {"IDIST_BUILD_GROWTH": "16"},                                   # narrow steps hold g / 16 insertions instead of g / 32
    "growth8": {"IDIST_BUILD_GROWTH": "8"},
    "growth16_check": {"IDIST_BUILD_GROWTH": "16", "IDIST_BUILD_CHECK": "1"},
    "regs512": {"IDIST_BUILD_A_REGS": "512"},                                   # descents: one 512-register wave per SIMD, four per CU
    "regs512_w3": {"IDIST_BUILD_A_REGS": "512", "IDIST_BUILD_A_WAVES": "3"},
    # (the session of commit d4e16c2 — profiles/probe_r04b_build_schedule.jsonl — had the layout under two knobs:
    #  r03_schedule = streams_off, default = streams_all, two_descent_streams_only / selection_stream_only = one of the two)
}
names = sys.argv[2].split(",") if len(sys.argv) > 2 else list(CASES)
truth = None
for nm in names:
    env = CASES[nm]
    os.environ.update(env)
    try:
        secs = []
        for _ in range(reps):
            h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder().max_batch(int(env.get("PB_MAX_BATCH", 0))))
            st = h.build_stats()
            secs.append(round(st.seconds, 4))
        if truth is None:
            truth, _ = h.bruteforce(q, 10)
        got = h.search_batch(q, ida.Search())
        rec = float(np.mean([len(set(got.pid[i, :10].tolist()) & set(truth[i].tolist())) / 10 for i in range(len(q))]))
        zero, _ = h.into_parts()
        ab = int(st.n_dist * 4 * dim + st.n_exp0 * 256 + st.n_expU * 128 + st.n_heur_rows * 4 * dim + st.n_updates * 512 + n * 256)
        row = dict(probe="build_schedule", commit=bench.source_stamp(), case=nm, n=n, dim=dim, env=env, seconds=secs, best=min(secs),
                   frac_of_8TBps=round(ab / min(secs) / 8e12, 4), recall_at_10=round(rec, 4), n_dist=int(st.n_dist), n_sel_pairs=int(st.n_sel_pairs),
                   n_heur_rows=int(st.n_heur_rows), n_updates=int(st.n_updates), n_updates_full=int(st.n_updates_full), batches=int(st.n_batches),
                   graph_checksum=int(zero.astype(np.uint64).sum()))
        del h, zero
    except Exception as e:  # noqa: BLE001
        row = dict(case=nm, env=env, err=repr(e)[:300])
    for k in env:
        os.environ.pop(k)
    print(json.dumps(row), flush=True)
    fo.write(json.dumps(row) + "\n")
    fo.flush()
