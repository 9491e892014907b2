"""Build seconds / work counters / recall@10 / graph checksum under schedule knobs, one process, one data set, every case PB_REPS times.
The knobs exist in the TEST build only (libidist_variants.so: the product's sources + knobs), which this script loads.
usage: python scripts/probe_build_knobs.py out.jsonl "name:KEY=VAL,KEY=VAL" "name2:..." ...   (PB_N / PB_DIM / PB_REPS; "default:" = no knob)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import instant_distance_amd as ida  # noqa: E402
from instant_distance_amd import _capi  # noqa: E402

torch.cuda.init()
_capi._singleton = _capi.Lib(os.path.join(os.path.dirname(_capi.LIB_PATH), os.environ.get("PB_LIB", "libidist_variants.so")))
fo = open(sys.argv[1], "a")
dev = torch.device("cuda", 0)
n, dim = int(os.environ.get("PB_N", 1_000_000)), int(os.environ.get("PB_DIM", 300))
reps = int(os.environ.get("PB_REPS", 3))
d_pts = bench.synth(torch, n, dim, 123456789, dev)
q = bench.synth(torch, 2000, dim, 123456790, dev).cpu().numpy()
torch.cuda.synchronize()
truth = None
for spec in sys.argv[2:]:
    nm, _, kv = spec.partition(":")
    env = dict(x.split("=", 1) for x in kv.split(",") if x)
    mb = int(env.pop("PB_MAX_BATCH", 0))
    os.environ.update(env)
    try:
        secs = []
        for _ in range(reps):
            h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder().max_batch(mb))
            st = h.build_stats()
            secs.append(round(st.seconds, 4))
        if truth is None:
            truth, _ = h.bruteforce(q, 10)
        got = h.search_batch(q, ida.Search())
        rec = float(np.mean([len(set(got.pid[i, :10].tolist()) & set(truth[i].tolist())) / 10 for i in range(len(q))]))
        zero, _ = h.into_parts()
        ab = int(st.n_dist * 4 * dim + st.n_exp0 * 256 + st.n_expU * 128 + st.n_heur_rows * 4 * dim + st.n_updates * 512 + n * 256)
        row = dict(probe="build_knobs", commit=bench.source_stamp(), case=nm, n=n, dim=dim, env=env, max_batch=mb, seconds=secs, best=min(secs),
                   frac_of_8TBps=round(ab / min(secs) / 8e12, 4), recall_at_10=round(rec, 4), n_dist=int(st.n_dist), n_sel_pairs=int(st.n_sel_pairs),
                   n_heur_rows=int(st.n_heur_rows), n_updates=int(st.n_updates), n_updates_full=int(st.n_updates_full), batches=int(st.n_batches),
                   graph_checksum=int(zero.astype(np.uint64).sum()))
        del h, zero
    except Exception as e:  # noqa: BLE001
        row = dict(case=nm, env=env, err=repr(e)[:300])
    for k in env:
        os.environ.pop(k, None)
    print(json.dumps(row), flush=True)
    fo.write(json.dumps(row) + "\n")
    fo.flush()
