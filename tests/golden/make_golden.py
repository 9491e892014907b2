"""Regenerates tests/golden/*.npz from the CPU oracle (run: python tests/golden/make_golden.py).

The reference (Rust) cannot be run in this image, so these are NOT outputs of the reference itself;
they freeze the behaviour of the oracle as pinned by tests/test_oracle_golden.py (the reference's own
known answers), so that any later change to the oracle or to the GPU engine shows up as a diff:
seeded inputs -> the full zero-layer adjacency, layer snapshots, search ids / distance bits / work
counters for the two shipped distances and both neighbour-selection modes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (n, dim, nq, metric, has_heuristic, data kind, ef_search)
    "l2sq_300d_heuristic": (400, 300, 12, po.METRIC_L2SQ, 1, "uniform", 100),
    "l2_2d_heuristic": (500, 2, 12, po.METRIC_L2, 1, "uniform", 100),
    "l2_3d_grid_heuristic": (300, 3, 12, po.METRIC_L2, 1, "grid", 20),
    "l2sq_16d_simple": (450, 16, 12, po.METRIC_L2SQ, 0, "uniform", 64),
}


def inputs(name):
    n, dim, nq, metric, heur, kind, ef = CASES[name]
    rng = np.random.default_rng(sum(map(ord, name)))
    if kind == "grid":
        pts = rng.integers(0, 6, size=(n, dim)).astype(np.float32)
        q = rng.integers(0, 6, size=(nq, dim)).astype(np.float32)
    else:
        pts = rng.random((n, dim), dtype=np.float32)
        q = rng.random((nq, dim), dtype=np.float32)
    return pts, q


def main():
    for name, (n, dim, nq, metric, heur, kind, ef) in CASES.items():
        pts, q = inputs(name)
        cfg = po.default_config(metric=metric, has_heuristic=heur, ef_search=ef)
        ix = po.Index.build(pts, cfg, threads=1)
        res = ix.search(q)
        layers = ix.layers
        np.savez_compressed(os.path.join(HERE, name + ".npz"), points=pts, queries=q, zero=ix.zero,
                            n_layers=np.array(len(layers)), **{f"layer{i}": l for i, l in enumerate(layers)},
                            pid=res.pid, dist_bits=res.dist.view(np.uint32), count=res.count, counters=res.counters,
                            meta=np.array([n, dim, nq, metric, heur, ef]))
        print(name, "written")


if __name__ == "__main__":
    main()
