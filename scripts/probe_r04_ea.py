"""Partial-distance early abandon on the headline kernel: C3 index, 10k queries, ef_search = 100, the
wide on-chip walk with the id set, IDIST_EA = 0 (off) / 4..7 blocks of the 9.5-block row fetched before the test.  Same box,
same index, same queries; results must be identical byte for byte (ids, distance bits, counts, work counters).
usage: python scripts/probe_r04_ea.py [out.jsonl]      (GPU box; needs instant-distance_amd/csrc/libidist_probe.so = `make probe`)
Run it under `rocprofv3 --pmc FETCH_SIZE` for the bytes per launch of every variant (kernel names carry the walk code)."""
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
out = open(sys.argv[1], "a") if len(sys.argv) > 1 else sys.stdout
dev = torch.device("cuda", 0)
n, dim, nq, ef = 1_000_000, 300, 10_000, int(os.environ.get("PB_EF", 100))
d_pts = bench.synth(torch, n, dim, 123456789, dev)
d_q = bench.synth(torch, nq, dim, 123456790, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
h.set_ef_search(ef)


def run(ea, reps=8):
    os.environ["IDIST_EA"] = str(ea)
    os.environ["IDIST_TAB_FORMAT"] = "ids"          # the set form the policy picks at ef 100 (the EA instantiations exist for it)
    s = ida.Search()
    o = (torch.empty(nq, ef, dtype=torch.int32, device=dev), torch.empty(nq, ef, dtype=torch.float32, device=dev),
         torch.empty(nq, dtype=torch.int32, device=dev), torch.empty(nq, 3, dtype=torch.int32, device=dev))
    for _ in range(reps + 2):
        h.search_batch_device(s, d_q.data_ptr(), nq, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    s.check_status()
    return s.kernel_times_ms(reps), [t.cpu().numpy() for t in o]


base_t, base_o = run(0)
ctr = base_o[3].astype(np.int64)
alg = int((ctr[:, 0] * 4 * dim + ctr[:, 1] * 256 + ctr[:, 2] * 128 + 8 * ef).sum())
for ea in (0, 4, 5, 6, 7, 0):
    t, o = run(ea)
    same = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(o, base_o))
    rec = {"probe": "early_abandon", "commit": bench.source_stamp(), "ef_search": ef, "ea_blocks": ea, "kernel_ms_mean": round(float(t.mean()), 4),
           "kernel_ms_min": round(float(t.min()), 4), "alg_GBps": round(alg / (float(t.mean()) * 1e-3) / 1e9, 1),
           "frac_of_8TBps_on_algorithmic_bytes": round(alg / (float(t.mean()) * 1e-3) / 8e12, 4), "identical_to_ea0": bool(same)}
    print(json.dumps(rec), file=out, flush=True)
