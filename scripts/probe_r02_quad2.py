"""Round-2 GPU probe, single-query latency on C3 (1M x 300, ef_search = 100), part 2: (a) wall time of the reference's
call pattern (one host query per call, a different query each time) with the kernel time of the very same launches,
both orders; (b) device-pointer launches of DISTINCT queries (cache-cold, unlike repeating one query); (c) with the
instrumented library (make phases; argv[2] = "phases") where a walk's time goes: before / in / after the distance passes.
usage: python scripts/probe_r02_quad2.py out.jsonl [phases]   (GPU box)"""
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
from instant_distance_amd import _capi  # noqa: E402

out_path = sys.argv[1]
phases = len(sys.argv) > 2 and sys.argv[2].startswith("phases")
if phases:      # "phases": before / in / after the distance passes; "phases2": pop+peek+adjacency / visited set+compaction / push
    _capi._singleton = _capi.Lib(os.path.join(ROOT, "instant-distance_amd", "csrc", "libidist_" + sys.argv[2] + ".so"))
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
emit(what="build", seconds=h.build_stats().seconds, phases=sys.argv[2] if phases else None)
q_host = d_q[:512].cpu().numpy()
VARIANTS = (("single wave", {"IDIST_QUAD_NQ": "0"}), ("four waves", {"IDIST_QUAD_NQ": "4000000000"}))


def ctx(env):
    os.environ.pop("IDIST_QUAD_NQ", None)
    os.environ.update(env)
    s = ida.Search()
    h.search_batch(q_host[:1], s)        # binds the context while the knobs are set
    return s


if not phases:
    # (a) host-pointer calls, a different query each time
    for order in (VARIANTS, VARIANTS[::-1]):
        for nm, env in order:
            s = ctx(env)
            t0 = time.perf_counter()
            for i in range(128):
                h.search_batch(q_host[i:i + 1], s)
            wall = (time.perf_counter() - t0) / 128 * 1e3
            kt = s.kernel_times_ms(64)
            emit(what="host-pointer call per query", variant=nm, wall_ms=round(wall, 4), kernel_ms_median=round(float(np.median(kt)), 4),
                 kernel_ms_min=round(float(kt.min()), 4), kernel_ms_max=round(float(kt.max()), 4))
            # the same query again and again (what probe_r02_quad.py timed): cache-warm
            for i in range(64):
                h.search_batch(q_host[:1], s)
            emit(what="host-pointer call, one query repeated", variant=nm, kernel_ms_median=round(float(np.median(s.kernel_times_ms(48))), 4))

# (b) / (c) device pointers, distinct queries, one launch per query
o = (torch.empty(1, 100, dtype=torch.int32, device=dev), torch.empty(1, 100, dtype=torch.float32, device=dev),
     torch.empty(1, dtype=torch.int32, device=dev), torch.zeros(512, 3, dtype=torch.int32, device=dev))
for nm, env in VARIANTS:
    s = ctx(env)
    for i in range(256):
        h.search_batch_device(s, d_q[1000 + i:].data_ptr(), 1, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(),
                              o[3][i:].data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    kt = s.kernel_times_ms(64)
    row = {"what": "device-pointer launches, distinct queries", "variant": nm, "kernel_ms_median": round(float(np.median(kt)), 4),
           "kernel_ms_p10": round(float(np.percentile(kt, 10)), 4), "kernel_ms_p90": round(float(np.percentile(kt, 90)), 4)}
    if phases:
        c = o[3][:256].cpu().numpy().astype(np.float64) * 1e-5      # 10-ns ticks -> ms
        row.update(pre_ms=round(float(c[:, 0].mean()), 4), dist_ms=round(float(c[:, 1].mean()), 4), post_ms=round(float(c[:, 2].mean()), 4))
    emit(**row)

if phases:      # the same shares inside a full 10k-query batch (one wave per SIMD, the chip loaded)
    os.environ.pop("IDIST_QUAD_NQ", None)
    ob = (torch.empty(nq, 100, dtype=torch.int32, device=dev), torch.empty(nq, 100, dtype=torch.float32, device=dev),
          torch.empty(nq, dtype=torch.int32, device=dev), torch.zeros(nq, 3, dtype=torch.int32, device=dev))
    s = ida.Search()
    for _ in range(3):
        h.search_batch_device(s, d_q.data_ptr(), nq, ob[0].data_ptr(), ob[1].data_ptr(), ob[2].data_ptr(), ob[3].data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    c = ob[3].cpu().numpy().astype(np.float64) * 1e-5
    emit(what="full batch, per query", kernel_ms=round(float(np.median(s.kernel_times_ms(2))), 3), a_ms=round(float(c[:, 0].mean()), 4),
         b_ms=round(float(c[:, 1].mean()), 4), c_ms=round(float(c[:, 2].mean()), 4))
