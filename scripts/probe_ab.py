"""Standard A/B line for a kernel change (GPU box, ~15 s): C3 build seconds, full 10k-query batch (median kernel ms,
algorithmic TB/s), single-query kernel time over distinct queries (four-wave and single-wave walk), one checksum of the
results so that two libraries can be compared.  usage: python scripts/probe_ab.py <tag> [out.jsonl] [lib.so]"""
import json
import os
import sys
import zlib

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

tag = sys.argv[1] if len(sys.argv) > 1 else "ab"
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "probe_ab.jsonl")
if len(sys.argv) > 3:
    _capi._singleton = _capi.Lib(os.path.join(ROOT, "instant-distance_amd", "csrc", sys.argv[3]))
os.makedirs(os.path.dirname(out_path), exist_ok=True)
dev = torch.device("cuda", 0)
n, dim, nq = 1_000_000, 300, 10_000
d_pts = bench.synth(torch, n, dim, 123456789, dev)
d_q = bench.synth(torch, nq, dim, 123456790, dev)
torch.cuda.synchronize()
h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
row = {"tag": tag, "build_s": round(h.build_stats().seconds, 4)}
o = (torch.empty(nq, 100, dtype=torch.int32, device=dev), torch.empty(nq, 100, dtype=torch.float32, device=dev),
     torch.empty(nq, dtype=torch.int32, device=dev), torch.zeros(nq, 3, dtype=torch.int32, device=dev))
st = torch.cuda.current_stream().cuda_stream
s = ida.Search()
for _ in range(9):
    h.search_batch_device(s, d_q.data_ptr(), nq, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
torch.cuda.synchronize()
s.check_status()
t = float(np.median(s.kernel_times_ms(7)))
ctr = o[3].cpu().numpy().astype(np.int64)
alg = int((ctr[:, 0] * 4 * dim + ctr[:, 1] * 256 + ctr[:, 2] * 128 + 8 * 100).sum())
row.update(batch_ms=round(t, 3), TBps=round(alg / t / 1e9, 3), qps=round(nq / t * 1e3),
           crc=zlib.crc32(o[0].cpu().numpy().tobytes()) ^ zlib.crc32(o[1].cpu().numpy().tobytes()) ^ zlib.crc32(ctr.tobytes()))
for nm, env in (("quad", {"IDIST_QUAD_NQ": "4000000000"}), ("single", {"IDIST_QUAD_NQ": "0"})):
    os.environ.update(env)
    s1 = ida.Search()
    for i in range(192):
        h.search_batch_device(s1, d_q[2000 + i:].data_ptr(), 1, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
    torch.cuda.synchronize()
    row["nq1_" + nm + "_ms"] = round(float(np.median(s1.kernel_times_ms(64))), 4)
    os.environ.pop("IDIST_QUAD_NQ")
    del s1
for w in (256, 1024):
    s2 = ida.Search()
    for i in range(8):
        h.search_batch_device(s2, d_q[i * w:].data_ptr(), w, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
    torch.cuda.synchronize()
    row["nq%d_ms" % w] = round(float(np.median(s2.kernel_times_ms(6))), 4)
    del s2
print(json.dumps(row), flush=True)
open(out_path, "a").write(json.dumps(row) + "\n")
