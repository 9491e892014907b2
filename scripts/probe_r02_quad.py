"""Round-2 GPU probe: narrow batches on C3 (1M x 300, ef_search = 100) — four waves per query (IDIST_QUAD_NQ) against
the single-wave on-chip walk, kernel time from the context's own HIP events, results compared bit for bit; plus the
full 10k batch and the build time with the wave-scope wave_sync (regression check against profiles/rocprof_summary_r02a).
usage: python scripts/probe_r02_quad.py [out.jsonl]   (GPU box)"""
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

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "probe_r02_quad.jsonl")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
fo = open(out_path, "a")


def emit(**kw):
    print(json.dumps(kw), flush=True)
    fo.write(json.dumps(kw) + "\n")
    fo.flush()


dev = torch.device("cuda", 0)
n, dim, nq = 1_000_000, 300, 10_000
d_pts = bench.synth(torch, n, dim, 123456789, dev)
d_q = bench.synth(torch, nq, dim, 123456790, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
emit(what="build", seconds=h.build_stats().seconds)


def run(env, width, reps=8, ef=100):
    for k in ("IDIST_QUAD_NQ", "IDIST_VISITED", "IDIST_WALK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    s = ida.Search()
    o = (torch.full((width, ef), -1, dtype=torch.int32, device=dev), torch.zeros(width, ef, dtype=torch.float32, device=dev),
         torch.zeros(width, dtype=torch.int32, device=dev), torch.zeros(width, 3, dtype=torch.int32, device=dev))
    for _ in range(reps + 2):
        h.search_batch_device(s, d_q.data_ptr(), width, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    s.check_status()
    kt = s.kernel_times_ms(reps)
    return float(np.median(kt)), o


for w in (1, 2, 8, 32, 64, 128, 256, 384, 512, 768, 1024, 2048):
    t1, o1 = run({"IDIST_QUAD_NQ": "0"}, w)
    t4, o4 = run({"IDIST_QUAD_NQ": "4000000000"}, w)
    same = bool(torch.equal(o1[0], o4[0]) and torch.equal(o1[1].view(torch.int32), o4[1].view(torch.int32))
                and torch.equal(o1[2], o4[2]) and torch.equal(o1[3], o4[3]))
    emit(what="narrow batch", nq=w, single_wave_ms=round(t1, 4), quad_ms=round(t4, 4), identical=same,
         single_qps=round(w / t1 * 1e3), quad_qps=round(w / t4 * 1e3))

# other ef_search values at nq = 1 (the set spills to the bitmap beyond ~7k visited ids)
for ef in (10, 50, 200, 400):
    h.set_ef_search(ef)
    t1, o1 = run({"IDIST_QUAD_NQ": "0"}, 1, ef=ef)
    t4, o4 = run({"IDIST_QUAD_NQ": "4000000000"}, 1, ef=ef)
    emit(what="nq=1 by ef_search", ef=ef, single_wave_ms=round(t1, 4), quad_ms=round(t4, 4),
         identical=bool(torch.equal(o1[0], o4[0]) and torch.equal(o1[3], o4[3])))
h.set_ef_search(100)

# the reference's call: one host query per Hnsw::search (host pointers in and out)
q_host = d_q[:64].cpu().numpy()
for env, nm in (({"IDIST_QUAD_NQ": "0"}, "single wave"), ({}, "default")):
    for k in ("IDIST_QUAD_NQ",):
        os.environ.pop(k, None)
    os.environ.update(env)
    s = ida.Search()
    h.search_batch(q_host[:1], s)
    t0 = time.perf_counter()
    for i in range(64):
        h.search_batch(q_host[i:i + 1], s)
    emit(what="Hnsw::search wall per call (host pointers)", variant=nm, ms=round((time.perf_counter() - t0) / 64 * 1e3, 4))
os.environ.pop("IDIST_QUAD_NQ", None)

# full batch, default walk: regression check of the wave-scope wave_sync
t, o = run({}, nq, reps=6)
ctr = o[3].cpu().numpy().astype(np.int64)
alg = int((ctr[:, 0] * 4 * dim + ctr[:, 1] * 256 + ctr[:, 2] * 128 + 8 * 100).sum())
emit(what="full batch (10k), default walk", ms=round(t, 3), TBps=round(alg / t / 1e9, 3))
