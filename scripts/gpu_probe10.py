"""GPU probe 10: Bloom filter size A/B (two extra builds of the library, made on the box)."""
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
for words, lg in ((1024, 10), (4096, 12)):
    out = os.path.join("/tmp", f"libidist_bloom{words}.so")
    subprocess.check_call(f"/opt/rocm/bin/hipcc {flags} -DIDIST_BLOOM_LOG2_WORDS={lg} -shared -o {out} {here}/idist_capi.hip", shell=True)
    libs[words] = out
libs[2048] = _capi.LIB_PATH
n, dim = 1_000_000, 300
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), 10000, dim, "lowrank")
for rnd in range(2):
    for words in (2048, 1024, 4096):
        _capi._singleton = _capi.Lib(libs[words])
        h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
        s = ida.Search()
        for _ in range(8):
            h.search_batch(q, s)
        ms = s.kernel_times_ms(6)
        print(json.dumps({"bloom_words": words, "round": rnd, "build_s": round(h.build_stats().seconds, 3),
                          "search_ms_min": round(float(ms.min()), 3), "search_ms_med": round(float(np.median(ms)), 3)}), flush=True)
        del h, s
