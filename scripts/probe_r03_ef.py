"""Search kernel time by ef_search under the visited-set variants of the on-chip walk (round 3): 16-bit quotient set
(8 ids per bucket, single ids overflow to the bitmap), full-id set (frozen at 7/8), bitmap + Bloom walk, and what the
policy picks — on C3 / C4 / C5-sized indexes, several fresh contexts per point to show the spread.
usage: python scripts/probe_r03_ef.py out.jsonl C3[,C4,C5] [ef,ef,...]"""
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
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["C3"]
efs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [100, 150, 200, 300, 400, 800]
os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
fo = open(out_path, "a")
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
VARIANTS = (("q16", {"IDIST_VISITED": "onchip", "IDIST_W2_EF": "4000000000"}), ("q16w2", {"IDIST_VISITED": "onchip", "IDIST_W2_EF": "0"}),
            ("ids", {"IDIST_VISITED": "onchip", "IDIST_TAB_FORMAT": "ids"}), ("bitmap", {"IDIST_VISITED": "bitmap"}), ("default", {}))
# (q16: one 512-register wave per SIMD; q16w2: two 256-register waves per SIMD, as many as the CU's LDS holds — round 4)
REPS = int(os.environ.get("PB_REPS", 3))

for name in which:
    c = bench.CONFIGS[name]
    n, dim, nq = c["n"], c["dim"], c["nq"]
    d_pts = bench.synth(torch, n, dim, 123456789, dev)
    d_q = bench.synth(torch, nq, dim, 123456790, dev)
    torch.cuda.synchronize()
    h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
    del d_pts
    for ef in efs:
        h.set_ef_search(ef)
        o = (torch.empty(nq, ef, dtype=torch.int32, device=dev), torch.empty(nq, ef, dtype=torch.float32, device=dev),
             torch.empty(nq, dtype=torch.int32, device=dev), torch.zeros(nq, 3, dtype=torch.int32, device=dev))
        row = {"commit": bench.source_stamp(), "config": name, "n": n, "dim": dim, "nq": nq, "ef": ef}
        ref = None
        for nm, env in VARIANTS:
            os.environ.update(env)
            ms = []
            try:
                for _ in range(REPS):                       # fresh context each time: allocation placement included
                    s = ida.Search()
                    for _ in range(3):
                        h.search_batch_device(s, d_q.data_ptr(), nq, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
                    torch.cuda.synchronize()
                    s.check_status()
                    ms.append(round(float(np.median(s.kernel_times_ms(2))), 3))
                    del s
                row[nm + "_ms"] = ms
                chk = int(o[0].sum().item())
                if ref is None:
                    ref = chk
                row[nm + "_same_ids"] = chk == ref
            except Exception as e:  # noqa: BLE001
                row[nm + "_err"] = repr(e)[:160]
            for k in env:
                os.environ.pop(k)
        ctr = o[3].cpu().numpy().astype(np.int64)
        alg = int((ctr[:, 0] * 4 * dim + ctr[:, 1] * 256 + ctr[:, 2] * 128 + 8 * ef).sum())
        row["n_dist"] = round(float(ctr[:, 0].mean()), 1)
        for nm, _ in VARIANTS:
            if nm + "_ms" in row:
                row[nm + "_frac_of_8TBps"] = round(alg / (min(row[nm + "_ms"]) * 1e-3) / 8e12, 3)
        print(json.dumps(row), flush=True)
        fo.write(json.dumps(row) + "\n")
        fo.flush()
        del o
    del h, d_q
    torch.cuda.empty_cache()
