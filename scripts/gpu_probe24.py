"""GPU probe 24: does the PLACEMENT of the index in HBM matter?  Several imports of the same graph (same library,
same walk), searched launch by launch in turn; then the same with the walks swapped over the copies."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim, nq = 1_000_000, 300, 10000
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
q = gen(np.random.default_rng(123456790), nq, dim, "lowrank")
h0 = ida.Hnsw.from_ordered_points(pts, ida.Builder())
zero, layers = h0.into_parts()
copies = [("built in place", h0, ida.Search())]
for i in range(3):
    copies.append((f"import #{i + 1}", ida.Hnsw.from_parts(pts, zero, layers, ida.Builder()), ida.Search()))
for walk in ("overlap", "classic", "overlap", "classic"):
    os.environ.pop("IDIST_WALK", None)
    if walk == "classic":
        os.environ["IDIST_WALK"] = "classic"
    times = {name: [] for name, *_ in copies}
    for rep in range(10):
        for name, h, s in copies:
            h.search_batch(q, s)
            times[name].append(float(s.kernel_times_ms(1)[0]))
    print(json.dumps({"walk": walk, "ms_median_per_copy": {k: round(float(np.median(v[2:])), 3) for k, v in times.items()}}), flush=True)
# address of each copy's point buffer
from instant_distance_amd import _capi  # noqa: E402
import ctypes as C  # noqa: E402
for name, h, s in copies:
    b = _capi.DeviceBuffers()
    _capi.lib().check(_capi.lib().idist_index_device_buffers(h._h, C.byref(b)))
    print(json.dumps({"copy": name, "points_ptr": hex(b.points or 0), "zero_ptr": hex(b.zero or 0)}), flush=True)
