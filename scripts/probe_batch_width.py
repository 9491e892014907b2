"""Search kernel ms by batch width under knob cases (one process, one index): where the four-wave walk (narrow batches) hands over
to the one-wave-per-query walks.  Loads the TEST build.  usage: python scripts/probe_batch_width.py out.jsonl case [case ...]
(PB_N / PB_DIM / PB_EF / PB_WIDTHS; cases as "name:KEY=VAL,KEY=VAL", "default:" = no knob)"""
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
n, dim, ef = int(os.environ.get("PB_N", 1_000_000)), int(os.environ.get("PB_DIM", 300)), int(os.environ.get("PB_EF", 100))
widths = [int(x) for x in os.environ.get("PB_WIDTHS", "64,128,256,512,768,1024,2048").split(",")]
d_pts = bench.synth(torch, n, dim, 123456789, dev)
d_q = bench.synth(torch, 10_000, dim, 123456790, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
h.set_ef_search(ef)
st = torch.cuda.current_stream().cuda_stream
for w in widths:
    row = dict(probe="batch_width", commit=bench.source_stamp(), n=n, dim=dim, ef=ef, nq=w)
    o = (torch.empty(w, ef, dtype=torch.int32, device=dev), torch.empty(w, ef, dtype=torch.float32, device=dev),
         torch.empty(w, dtype=torch.int32, device=dev), torch.zeros(w, 3, dtype=torch.int32, device=dev))
    for spec in sys.argv[2:]:
        nm, _, kv = spec.partition(":")
        env = dict(x.split("=", 1) for x in kv.split(",") if x)
        os.environ.update(env)
        s = ida.Search()
        for i in range(9):
            h.search_batch_device(s, d_q[(i * w) % (10_000 - w):].data_ptr(), w, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
        torch.cuda.synchronize()
        s.check_status()
        row[nm + "_ms"] = round(float(np.median(s.kernel_times_ms(7))), 4)
        for k in env:
            os.environ.pop(k, None)
        del s
    print(json.dumps(row), flush=True)
    fo.write(json.dumps(row) + "\n")
    fo.flush()
