"""C3 build only (for profiling)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import instant_distance_amd as ida  # noqa: E402
from scripts.gpu_probe import gen  # noqa: E402

n, dim = 1_000_000, int(os.environ.get("BC_DIM", 300))
pts = gen(np.random.default_rng(123456789), n, dim, "lowrank")
h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
st = h.build_stats()
print("build_s", st.seconds, "n_sel_pairs", st.n_sel_pairs)
