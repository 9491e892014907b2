"""C3 (1M x 300): search kernel ms by ef_search for the on-chip walk (forced, IDIST_TAB_LOG2=13: the set spills to the bitmap
beyond ~7k visited ids) against the bitmap walk — the crossover decides kOnChipMaxEf.  usage: python scripts/probe_ef_paths.py out.jsonl"""
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

# the knobs this script steers with exist in the test build only (libidist_variants.so); PB_LIB names another library
torch.cuda.init()
_capi._singleton = _capi.Lib(os.path.join(ROOT, "instant-distance_amd", "csrc", os.environ.get("PB_LIB", "libidist_variants.so")))

fo = open(sys.argv[1], "a")
dev = torch.device("cuda", 0)
n, dim, nq = int(os.environ.get("PB_N", 1_000_000)), int(os.environ.get("PB_DIM", 300)), 10_000
d_pts = bench.synth(torch, n, dim, 123456789, dev)
d_q = bench.synth(torch, nq, dim, 123456790, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
st = torch.cuda.current_stream().cuda_stream
for ef in (100, 150, 200, 300, 400, 800):
    h.set_ef_search(ef)
    o = (torch.empty(nq, ef, dtype=torch.int32, device=dev), torch.empty(nq, ef, dtype=torch.float32, device=dev),
         torch.empty(nq, dtype=torch.int32, device=dev), torch.zeros(nq, 3, dtype=torch.int32, device=dev))
    row = {"n": n, "dim": dim, "ef": ef}
    for nm, env in (("on_chip_forced", {"IDIST_TAB_LOG2": "13"}), ("bitmap", {"IDIST_VISITED": "bitmap"}), ("default", {})):
        os.environ.update(env)
        s = ida.Search()
        try:
            for _ in range(4):
                h.search_batch_device(s, d_q.data_ptr(), nq, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
            torch.cuda.synchronize()
            s.check_status()
            row[nm + "_ms"] = round(float(np.median(s.kernel_times_ms(3))), 3)
        except Exception as e:  # noqa: BLE001
            row[nm + "_err"] = repr(e)[:120]
        for k in env:
            os.environ.pop(k)
        del s
    row["n_dist"] = round(float(o[3][:, 0].float().mean()), 1)
    print(json.dumps(row), flush=True)
    fo.write(json.dumps(row) + "\n")
    fo.flush()
