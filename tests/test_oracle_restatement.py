"""The C oracle (oracle/idist_oracle.c) against a SECOND restatement of the reference written separately from the Rust
source (tests/py_restatement.py): both must build the same graph, array for array, and answer queries with the same
ids in the same order.  The reference itself cannot run here (no rustc); its own known answers pin the oracle
behaviourally (tests/test_oracle_golden.py), this pins the two restatements to each other — including multi-layer
indexes (a large `ml` gives a few dozen points several layers) and `Heuristic::extend_candidates` without its locks."""
import numpy as np
import pytest

import py_restatement as pr


def _tables(oracle, pts, q, metric):
    n = len(pts)
    D = np.zeros((n, n), dtype=np.float64)
    for i in range(n):
        for j in range(i, n):
            D[i, j] = D[j, i] = float(oracle.distance(pts[i], pts[j], metric))      # symmetric bit for bit
    Q = np.array([[float(oracle.distance(qq, p, metric)) for p in pts] for qq in q])
    return D, Q


@pytest.mark.parametrize("n,dim,kw", [
    (1, 3, {}), (2, 3, {}), (45, 3, {}), (70, 2, {"metric": 1}), (90, 4, {"ml": 0.5}),
    (130, 3, {"ml": 0.7, "ef_construction": 20}), (110, 5, {"ml": 0.6, "keep_pruned": 0}),
    (80, 2, {"kind": "grid", "metric": 1, "ml": 0.6}),
    (60, 3, {"extend_candidates": 1}), (100, 4, {"extend_candidates": 1, "ml": 0.6, "ef_construction": 16}),
    (75, 2, {"extend_candidates": 1, "keep_pruned": 0, "metric": 1}),
    (85, 3, {"has_heuristic": 0}), (120, 2, {"has_heuristic": 0, "ml": 0.6, "metric": 1}),
    (90, 2, {"has_heuristic": 0, "kind": "grid", "metric": 1}),
])
def test_two_restatements_build_the_same_graph(oracle, n, dim, kw):
    kw = dict(kw)
    kind = kw.pop("kind", "uniform")
    rng = np.random.default_rng(1000 + n)
    pts = (rng.integers(0, 4, size=(n, dim)) if kind == "grid" else rng.random((n, dim))).astype(np.float32)
    q = rng.random((6, dim)).astype(np.float32) * (3 if kind == "grid" else 1)
    cfg = oracle.default_config(ef_search=25, **kw)
    oix = oracle.Index.build(pts, cfg, threads=1)
    D, Q = _tables(oracle, pts, q, cfg.metric)
    zero, layers = pr.build(D, n, cfg.ml, cfg.ef_construction, bool(cfg.extend_candidates), bool(cfg.keep_pruned),
                            bool(cfg.has_heuristic))
    assert np.array_equal(zero, oix.zero)
    assert len(layers) == len(oix.layers)
    for a, b in zip(layers, oix.layers):
        assert np.array_equal(a, b)
    want = oix.search(q)
    for i in range(len(q)):
        got = pr.search_index(zero, layers, lambda pid, i=i: float(Q[i][pid]), 25)
        c = int(want.count[i])
        assert len(got) == c
        assert [p for _, p in got] == want.pid[i, :c].tolist()
        assert np.array_equal(np.array([d for d, _ in got], dtype=np.float32).view(np.uint32), want.dist[i, :c].view(np.uint32))
