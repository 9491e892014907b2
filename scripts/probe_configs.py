"""The other BASELINE.json configurations with the current kernels (GPU box): C2 100k x 128 (uniform and fastText-shape),
C4 1M x 768 with 65,536 queries, C5 10M x 768 (one GPU's replica) — build seconds, search kernel ms at ef_search 100 and
200, recall@10 on a query sample against the exact ground truth, algorithmic HBM rate.
usage: python scripts/probe_configs.py out.jsonl C2,C4[,C5]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import instant_distance_amd as ida  # noqa: E402

out_path = sys.argv[1]
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["C2", "C4"]
os.makedirs(os.path.dirname(out_path), exist_ok=True)
fo = open(out_path, "a")
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream


def emit(**kw):
    print(json.dumps(kw), flush=True)
    fo.write(json.dumps(kw) + "\n")
    fo.flush()


def run(name, n, dim, nq, efs, uniform=False, gtq=1000):
    if uniform:
        g = torch.Generator(device=dev).manual_seed(123456789)
        d_pts = torch.rand(n, dim, generator=g, device=dev, dtype=torch.float32)
        d_q = torch.rand(nq, dim, generator=g, device=dev, dtype=torch.float32)
    else:
        d_pts = bench.synth(torch, n, dim, 123456789, dev)
        d_q = bench.synth(torch, nq, dim, 123456790, dev)
    torch.cuda.synchronize()
    h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
    bs = h.build_stats()
    row = {"config": name, "n": n, "dim": dim, "nq": nq, "data": "uniform" if uniform else "fastText-shape",
           "build_s": round(bs.seconds, 3), "build_points_per_s": round(n / bs.seconds)}
    truth, _ = h.bruteforce(d_q[:gtq].cpu().numpy(), 10)
    for ef in efs:
        h.set_ef_search(ef)
        o = (torch.empty(nq, ef, dtype=torch.int32, device=dev), torch.empty(nq, ef, dtype=torch.float32, device=dev),
             torch.empty(nq, dtype=torch.int32, device=dev), torch.zeros(nq, 3, dtype=torch.int32, device=dev))
        s = ida.Search()
        for _ in range(5):
            h.search_batch_device(s, d_q.data_ptr(), nq, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
        torch.cuda.synchronize()
        s.check_status()
        t = float(np.median(s.kernel_times_ms(3)))
        ctr = o[3].cpu().numpy().astype(np.int64)
        alg = int((ctr[:, 0] * 4 * dim + ctr[:, 1] * 256 + ctr[:, 2] * 128 + 8 * ef).sum())
        got = o[0][:gtq, :10].cpu().numpy().astype(np.uint32)
        rec = float(np.mean([len(set(got[i].tolist()) & set(truth[i].tolist())) / 10 for i in range(gtq)]))
        row["ef%d" % ef] = {"kernel_ms": round(t, 3), "qps": round(nq / t * 1e3), "recall_at_10": round(rec, 4),
                            "alg_TBps": round(alg / t / 1e9, 3), "frac_of_8TBps": round(alg / t / 1e9 / 8, 3),
                            "n_dist_per_query": round(float(ctr[:, 0].mean()), 1)}
        del s, o
    emit(**row)
    del h, d_pts, d_q
    torch.cuda.empty_cache()


if "C2" in which:
    run("C2", 100_000, 128, 10_000, (100,), uniform=True)
    run("C2", 100_000, 128, 10_000, (100,))
if "C4" in which:
    run("C4", 1_000_000, 768, 65_536, (100, 200))
if "C5" in which:
    run("C5 (one GPU's replica)", 10_000_000, 768, 65_536, (100, 200))
