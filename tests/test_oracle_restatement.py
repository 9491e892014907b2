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


try:
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st
except Exception:  # noqa: BLE001
    given = None

if given is not None:
    @settings(max_examples=150, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow],
              derandomize=True)
    @given(n=st.integers(1, 100), dim=st.integers(1, 5), ml=st.sampled_from([0.28853902, 0.45, 0.6, 0.75]),
           efc=st.sampled_from([1, 3, 8, 30, 100]), metric=st.integers(0, 1), mode=st.sampled_from(["heur", "heur-nokeep", "extend", "simple"]),
           lattice=st.booleans(), seed=st.integers(0, 10_000))
    def test_two_restatements_agree_on_drawn_configs(oracle, n, dim, ml, efc, metric, mode, lattice, seed):
        """The same comparison over drawn configurations (hypothesis, derandomised): sizes around the layer thresholds,
        tiny ef_construction (push refuses ranks >= ef, lib.rs:713), lattices with exact ties, every selection mode."""
        rng = np.random.default_rng(seed)
        pts = (rng.integers(0, 3, size=(n, dim)) if lattice else rng.random((n, dim))).astype(np.float32)
        cfg = oracle.default_config(ml=ml, ef_construction=efc, metric=metric, has_heuristic=int(mode != "simple"),
                                    extend_candidates=int(mode == "extend"), keep_pruned=int(mode != "heur-nokeep"))
        oix = oracle.Index.build(pts, cfg, threads=1)
        D, _ = _tables(oracle, pts, pts[:0], metric)
        zero, layers = pr.build(D, n, cfg.ml, efc, mode == "extend", mode != "heur-nokeep", mode != "simple")
        assert np.array_equal(zero, oix.zero)
        assert len(layers) == len(oix.layers) and all(np.array_equal(a, b) for a, b in zip(layers, oix.layers))
