"""PMC calibration for the gather access pattern (MI355X_MICROARCH.md §HBM: FETCH_SIZE must be
calibrated on a known byte count in your own access pattern).  Runs distance_batch_kernel over a
random permutation of ALL rows of a 1M x 300 index (1.216 GB >> 256 MiB Infinity Cache): every row
is read exactly once with the same 8-lanes-per-row float4 loads as the search kernel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402

n, dim = 1_000_000, 300
rng = np.random.default_rng(0)
pts = rng.random((n, dim), dtype=np.float32)
h = ida.Hnsw.from_parts(pts, np.full((n, 64), 0xFFFFFFFF, np.uint32), [], ida.Builder())
ids = rng.permutation(n).astype(np.uint32).reshape(1, n)
q = rng.random((1, dim), dtype=np.float32)
for _ in range(3):
    d = h.distances(q, ids)
info = h.info()
print("known_read_bytes_per_launch", n * info.row_stride * 4 + n * 4, "row_stride", info.row_stride, float(d[0, 0]))
