"""GPU probe 23: classic vs overlap walk (IDIST_WALK) at 128 / 300 / 768 dimensions, launch-by-launch interleaved,
plus build time under either walk; for 768-d also 2 rounds in flight (library variant built on the box)."""
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
alt = "/tmp/libidist_rif24x2.so"
subprocess.check_call(f"/opt/rocm/bin/hipcc {flags} -DIDIST_RIF24_OVERLAP=2 -shared -o {alt} {here}/idist_capi.hip", shell=True)
base_lib = _capi.Lib(_capi.LIB_PATH)
alt_lib = _capi.Lib(alt)
for n, dim, nqs in ((1_000_000, 300, (10000, 16384)), (100_000, 128, (10000,)), (1_000_000, 768, (16384,))):
    pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
    q = gen(np.random.default_rng(123456790), 16384, dim, "lowrank")
    builds = {}
    for rnd in range(2):
        for walk in ("overlap", "classic"):
            _capi._singleton = base_lib
            os.environ.pop("IDIST_WALK", None)
            if walk == "classic":
                os.environ["IDIST_WALK"] = "classic"
            h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
            builds.setdefault(walk, []).append(round(h.build_stats().seconds, 4))
            if rnd == 0 and walk == "overlap":
                zero, layers = h.into_parts()
            del h
    print(json.dumps({"dim": dim, "n": n, "build_s": builds}), flush=True)
    cfgs = [("overlap", base_lib, None), ("classic", base_lib, "classic")]
    if dim == 768:
        cfgs.append(("overlap, 2 rounds in flight", alt_lib, None))
    objs = []
    for name, lib, walk in cfgs:
        _capi._singleton = lib
        objs.append((name, lib, walk, ida.Hnsw.from_parts(pts, zero, layers, ida.Builder()), ida.Search()))
    for nq in nqs:
        times = {name: [] for name, *_ in objs}
        ref = None
        for rep in range(12):
            for name, lib, walk, h, s in objs:
                _capi._singleton = lib
                os.environ.pop("IDIST_WALK", None)
                if walk:
                    os.environ["IDIST_WALK"] = walk
                r = h.search_batch(q[:nq], s)
                times[name].append(float(s.kernel_times_ms(1)[0]))
                if ref is None:
                    ref = r.pid
                assert np.array_equal(r.pid, ref)
        for name, t in times.items():
            t = np.array(t[2:])
            print(json.dumps({"dim": dim, "walk": name, "nq": nq, "ms_median": round(float(np.median(t)), 3), "ms_min": round(float(t.min()), 3),
                              "qps_median": round(nq / float(np.median(t)) * 1e3)}), flush=True)
    del objs
