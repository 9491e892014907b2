"""GPU probe 15: do the latency variant's overlaps pay in the throughput variant too?  Library variants built on
the box: adjacency prefetch (IDIST_TP_PFA) and visited-byte overlap (IDIST_TP_OVL), full batches."""
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
libs = {"base": _capi.LIB_PATH}
for name, d in (("pfa", "-DIDIST_TP_PFA=1"), ("ovl", "-DIDIST_TP_OVL=1"), ("pfa+ovl", "-DIDIST_TP_PFA=1 -DIDIST_TP_OVL=1")):
    out = os.path.join("/tmp", f"libidist_{name.replace('+', '_')}.so")
    subprocess.check_call(f"/opt/rocm/bin/hipcc {flags} {d} -shared -o {out} {here}/idist_capi.hip", shell=True)
    libs[name] = out
n, dim = 1_000_000, 300
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), 10000, dim, "lowrank")
h0 = ida.Hnsw.from_ordered_points(pts, ida.Builder())
zero, layers = h0.into_parts()
ref = h0.search_batch(q, ida.Search()).pid
del h0
for rnd in range(2):
    for name in ("base", "pfa", "ovl", "pfa+ovl"):
        _capi._singleton = _capi.Lib(libs[name])
        h = ida.Hnsw.from_parts(pts, zero, layers, ida.Builder())
        s = ida.Search()
        for _ in range(8):
            r = h.search_batch(q, s)
        ms = s.kernel_times_ms(6)
        hb = ida.Hnsw.from_ordered_points(pts, ida.Builder())
        print(json.dumps({"variant": name, "round": rnd, "search_ms_min": round(float(ms.min()), 3), "search_ms_med": round(float(np.median(ms)), 3),
                          "same_ids": bool(np.array_equal(r.pid, ref)), "build_s": round(hb.build_stats().seconds, 3)}), flush=True)
        del h, s, hb
