"""Seeded fuzz of the walk's reject filter on the GPU (product library, default policy): random dimension / size / distribution / metric /
ef_search; the GPU builds the index (default schedule: filtered descents where the policy has them), the oracle searches the exported
graph — ids, order, counts, distance bits and work counters of a wide batch must be identical, the filter must have been at work
wherever it applies; then a small exact build (max_batch = 1, descents WITH the filter at every row length, test build) must be
byte-identical to the oracle's.  Needs the oracle (test infrastructure).  FUZZ_SEED / FUZZ_COUNT."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
import instant_distance_amd as ida  # noqa: E402
from test_filter import _data  # noqa: E402

oracle.build_lib()
oracle.lib()
seed, count = int(os.environ.get("FUZZ_SEED", 6000)), int(os.environ.get("FUZZ_COUNT", 30))
DIMS = (24, 64, 100, 128, 200, 300, 384, 512, 640, 768, 1024, 1500)
KINDS = ("lowrank", "uniform", "offset", "outliers", "heavytail", "grid")
bad = 0
for i in range(count):
    rng = np.random.default_rng(seed + i)
    dim, kind = int(rng.choice(DIMS)), str(rng.choice(KINDS))
    if kind == "grid":
        dim = int(rng.choice((6, 8, 12)))
    n = int(rng.integers(20_000, 60_000)) if dim <= 512 else int(rng.integers(8_000, 20_000))
    if i % 2 and dim < 256:                          # short rows take the filtered walk beyond the L2's reach (32 MB of rows) only
        n = max(n, int(48e6 / (4 * max(dim, 16))))
    metric, ef = int(rng.integers(0, 2)), int(rng.choice((10, 50, 100, 200, 400)))
    cfg = dict(case=i, n=n, dim=dim, kind=kind, metric=metric, ef=ef)
    try:
        pts = _data(rng, kind, n, dim)
        q = _data(rng, kind, 600, dim)
        q[0] = pts[n // 3]
        b = ida.Builder().metric(metric).ef_search(ef)
        h = ida.Hnsw.from_ordered_points(pts, b)
        zero, layers = h.into_parts()
        oix = oracle.Index.from_arrays(pts, zero, layers, oracle.default_config(metric=metric, ef_search=ef))
        want = oix.search(q, threads=16)
        s = ida.Search()
        got = h.search_batch(q, s, counters=True)
        pc.check_search_result(got, want)
        seen, rej = s.filter_counts()
        st = h.build_stats()
        cfg.update(filter_examined=seen, filter_rejected_share=round(rej / seen, 3) if seen else None,
                   build_filter_rejected_share=round(st.n_filter_rejected / st.n_filter_examined, 3) if st.n_filter_examined else None)
        # exact build, filtered descents forced
        m = min(n, 2500 if dim <= 640 else 1200)
        oex = oracle.Index.build(pts[:m], oracle.default_config(metric=metric, ef_construction=60), threads=1)
        with pc.search_variant({"IDIST_BUILD_QUAD": "0", "IDIST_BUILD_FILTER": "1"}):
            hx = ida.Hnsw.from_ordered_points(pts[:m], ida.Builder().metric(metric).ef_construction(60).max_batch(1))
            zx, lx = hx.into_parts()
        assert np.array_equal(zx, oex.zero) and all(np.array_equal(x, y) for x, y in zip(lx, oex.layers)), "exact build differs"
        print(json.dumps(cfg), flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        cfg["error"] = repr(e)[:300]
        print(json.dumps(cfg), flush=True)
print(json.dumps({"fuzz_seed": seed, "cases": count, "failed": bad}), flush=True)
sys.exit(1 if bad else 0)
