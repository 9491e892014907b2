"""GPU probe 12: rounds in flight of the latency variant (dist_rounds_inflight) at 300-d: library variants
built on the box with -DIDIST_RIF9=3/4/8; narrow-batch search latency and build time."""
import json
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from instant_distance_amd import _capi  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

here = os.path.dirname(_capi.LIB_PATH)
flags = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "
         "-fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value -Wno-unused-result")
libs = {}
for rif in (3, 4, 8):
    out = os.path.join("/tmp", f"libidist_rif{rif}.so")
    subprocess.check_call(f"/opt/rocm/bin/hipcc {flags} -DIDIST_RIF9={rif} -shared -o {out} {here}/idist_capi.hip", shell=True)
    libs[rif] = out
n, dim = 1_000_000, 300
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), 4096, dim, "lowrank")
ref = None
for rif in (3, 4, 8):
    _capi._singleton = _capi.Lib(libs[rif])
    h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
    print(json.dumps({"rif": rif, "build_s": round(h.build_stats().seconds, 3)}), flush=True)
    s = ida.Search()
    for nq in (1, 32, 256, 512, 1024, 2048):
        reps = 30
        for i in range(reps):
            r = h.search_batch(q[(i * 7) % (4096 - nq + 1):][:nq], s)
        ms = s.kernel_times_ms(reps - 2)
        print(json.dumps({"rif": rif, "nq": nq, "kernel_ms_med": round(float(np.median(ms)), 4), "kernel_ms_min": round(float(ms.min()), 4),
                          "kernel_qps": round(nq / (float(np.median(ms)) * 1e-3))}), flush=True)
    r = h.search_batch(q[:1024], s)
    if ref is None:
        ref = r
    else:
        print(json.dumps({"rif": rif, "same_ids_as_first_variant": bool(np.array_equal(r.pid, ref.pid))}), flush=True)
    del h, s
