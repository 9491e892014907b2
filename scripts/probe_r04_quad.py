"""Where a one-query call spends its zero-layer walk: the four-wave walk's leader and helpers, 10-ns
ticks per segment summed over 256 calls on the C3 index (1M x 300-d, ef_search 100 / 400), from the measurement build
(`make probe`: g_quad_probe in idist_device.hpp, read through idist_probe_quad).
usage: python scripts/probe_r04_quad.py [out.jsonl]      (GPU box; needs instant-distance_amd/csrc/libidist_probe.so)"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import instant_distance_amd as ida  # noqa: E402
from instant_distance_amd import _capi  # noqa: E402

_capi._singleton = _capi.Lib(os.path.join(ROOT, "instant-distance_amd", "csrc", "libidist_probe.so"))
raw = ctypes.CDLL(os.path.join(ROOT, "instant-distance_amd", "csrc", "libidist_probe.so"))
raw.idist_probe_quad.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = open(sys.argv[1], "a") if len(sys.argv) > 1 else sys.stdout
dev = torch.device("cuda", 0)
n, dim, calls = 1_000_000, int(os.environ.get("PB_DIM", 300)), 256
d_pts = bench.synth(torch, n, dim, 123456789, dev)
d_q = bench.synth(torch, calls, dim, 123456790, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())


def read(reset=1):
    buf = (ctypes.c_ulonglong * 24)()
    assert raw.idist_probe_quad(buf, reset) == 0
    return np.array(list(buf), dtype=np.float64)


for ef in (100, 400):
    h.set_ef_search(ef)
    s = ida.Search()
    o = (torch.empty(1, ef, dtype=torch.int32, device=dev), torch.empty(1, ef, dtype=torch.float32, device=dev),
         torch.empty(1, dtype=torch.int32, device=dev), torch.empty(1, 3, dtype=torch.int32, device=dev))
    stream = torch.cuda.current_stream().cuda_stream
    for i in range(8):
        h.search_batch_device(s, d_q.data_ptr() + 4 * dim * i, 1, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), stream)
    read()
    times = []
    for i in range(calls):
        h.search_batch_device(s, d_q.data_ptr() + 4 * dim * i, 1, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), stream)
        torch.cuda.synchronize()
        times.append(float(s.kernel_times_ms(1)[-1]))
    s.check_status()
    p = read()
    nexp = max(p[6], 1.0)
    ns = lambda v: round(float(v) * 10.0 / nexp, 1)          # ticks are 10 ns
    rec = {"probe": "quad_walk_segments", "commit": bench.source_stamp(), "dim": dim, "ef_search": ef, "calls": calls,
           "kernel_ms_median": round(float(np.median(times)), 4),
           "expansions_per_call": round(nexp / calls, 1), "new_ids_per_expansion": round(p[10] / nexp, 1),
           "asked": round(p[9] / nexp, 3), "taken": round(p[7] / nexp, 3), "aborted": round(p[8] / nexp, 4),
           "leader_ns_per_expansion": {"pop": ns(p[0]), "adjacency_and_peeks": ns(p[1]), "wait_for_helpers": ns(p[2]),
                                       "inserts_after_take": ns(p[11]), "own_pass_when_not_taken": ns(p[3]), "post_request": ns(p[4]),
                                       "push_and_truncate": ns(p[5]), "sum": ns(p[0] + p[1] + p[2] + p[11] + p[3] + p[4] + p[5])},
           "helper_ns_per_request": {f"wave{w + 1}": {"lookup": round(p[12 + 3 * w] * 10 / max(p[14 + 3 * w], 1), 1),
                                                     "rows": round(p[13 + 3 * w] * 10 / max(p[14 + 3 * w], 1), 1),
                                                     "requests": int(p[14 + 3 * w])} for w in range(3)}}
    print(json.dumps(rec), file=out, flush=True)
