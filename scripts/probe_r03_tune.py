"""C3 search (10k queries) through the tuning build's instantiations of the on-chip walk (make -C instant-distance_amd/csrc tune):
one wave per SIMD with 512 registers vs two with 256, rounds in flight, 32-KB vs 16-KB quotient set, 4 vs 8 waves per CU.
usage: python scripts/probe_r03_tune.py out.jsonl [ef,ef,...]"""
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

_capi._singleton = _capi.Lib(os.path.join(ROOT, "instant-distance_amd", "csrc", "libidist_tune.so"))
fo = open(sys.argv[1], "a")
efs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [100, 200, 400]
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
n, dim, nq = 1_000_000, 300, 10_000
d_pts = bench.synth(torch, n, dim, 123456789, dev)
d_q = bench.synth(torch, nq, dim, 123456790, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
# (tune case, set size log2, waves per CU)
CASES = [(10, 13, 4), (11, 13, 4), (11, 12, 8), (12, 12, 8), (13, 12, 8), (14, 12, 8), (11, 12, 6), (10, 12, 4)]
for ef in efs:
    h.set_ef_search(ef)
    o = (torch.empty(nq, ef, dtype=torch.int32, device=dev), torch.empty(nq, ef, dtype=torch.float32, device=dev),
         torch.empty(nq, dtype=torch.int32, device=dev), torch.zeros(nq, 3, dtype=torch.int32, device=dev))
    ref = None
    for tune, tl2, wpc in CASES:
        env = {"IDIST_TUNE": str(tune), "IDIST_TAB_LOG2": str(tl2), "IDIST_WAVES_PER_CU": str(wpc)}
        os.environ.update(env)
        row = {"ef": ef, "tune": tune, "tab_log2": tl2, "waves_per_cu": wpc}
        try:
            s = ida.Search()
            for _ in range(4):
                h.search_batch_device(s, d_q.data_ptr(), nq, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
            torch.cuda.synchronize()
            s.check_status()
            row["kernel_ms"] = round(float(np.median(s.kernel_times_ms(3))), 3)
            chk = int(o[0].sum().item())
            ref = chk if ref is None else ref
            row["same_ids"] = chk == ref
            del s
        except Exception as e:  # noqa: BLE001
            row["err"] = repr(e)[:160]
        for k in env:
            os.environ.pop(k)
        print(json.dumps(row), flush=True)
        fo.write(json.dumps(row) + "\n")
        fo.flush()
