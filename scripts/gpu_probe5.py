"""GPU probe 5: A/B of the non-temporal row-load variant (libidist_nt.so) vs the default build."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from instant_distance_amd import _capi  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim = 1_000_000, 300
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), 10000, dim, "lowrank")
here = os.path.dirname(_capi.LIB_PATH)
res = {}
for rnd in range(2):
    for name in ("libidist.so", "libidist_nt.so"):
        _capi._singleton = _capi.Lib(os.path.join(here, name))
        h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
        st = h.build_stats()
        s = ida.Search()
        for _ in range(6):
            r = h.search_batch(q, s)
        ms = s.kernel_times_ms(5)
        print(json.dumps({"lib": name, "round": rnd, "build_s": round(st.seconds, 3), "search_ms_min": round(float(ms.min()), 3),
                          "search_ms_med": round(float(np.median(ms)), 3)}), flush=True)
        del h, s
