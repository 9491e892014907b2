"""The walk's reject filter (FilterView / filter_rounds in csrc/idist_device.hpp): `Search::push` (core/lib.rs:704-720) needs the
distance of a candidate only when it accepts it (`Err(idx) if idx < ef`); the others need a PROOF that their canonical distance
exceeds nearest[ef-1]'s, which a one-byte-per-coordinate copy of the row can give.  Whatever the filter does, ids, order, counts,
distance bits and the work counters {n_dist, n_exp0, n_expU} must be the oracle's — and the filter must actually be at work
(idist_search_ctx_filter_counts says what it looked at and what it rejected)."""
import numpy as np
import pytest

import parity_cases as pc
from engines import engine_params

ON_CHIP = {"IDIST_QUAD_NQ": "0", "IDIST_VISITED": "onchip"}          # wide on-chip walk whatever the index size (test-build knobs)
WALKS = (("two thin waves per SIMD on the quotient set (the filtered wide walk)", dict(ON_CHIP)),
         ("the same with a tiny quotient set: ids overflow to the bitmap (second pass)", {**ON_CHIP, "IDIST_TAB_LOG2": "7"}),
         ("... and a mid-sized one", {**ON_CHIP, "IDIST_TAB_LOG2": "10"}))


@pytest.fixture(params=engine_params())
def eng(request, engine_loader):
    return engine_loader(request.param), request.param


def S(kind, emu, gpu):
    return emu if kind == "emu" else gpu


def _data(rng, kind_, n, dim):
    if kind_ == "outliers":              # a few coordinates far outside the lattice: clamped, their error is in the row's record
        a = pc.gen_points(rng, n, dim, "lowrank")
        rows = rng.choice(n, size=max(2, n // 25), replace=False)
        a[rows, rng.integers(0, dim, size=len(rows))] *= np.float32(40.0)
        return a
    if kind_ == "offset":                # data far from the origin: the padding coordinates (zeros) must not count
        return (pc.gen_points(rng, n, dim, "uniform") + np.float32(7.0)).astype(np.float32)
    if kind_ == "heavytail":             # Cauchy coordinates: the lattice covers the bulk, every row has clamped outliers -> the filter
        return rng.standard_cauchy((n, dim)).astype(np.float32)   # rejects little and a walk switches it off for itself
    if kind_ == "constant":              # degenerate range
        return np.full((n, dim), 0.5, dtype=np.float32)
    return pc.gen_points(rng, n, dim, kind_)


@pytest.mark.parametrize("dim,kind_,metric", [(128, "lowrank", 0), (300, "lowrank", 0), (300, "uniform", 1), (16, "uniform", 0),
                                              (100, "lowrank", 1), (7, "grid", 0), (48, "outliers", 0), (33, "offset", 0),
                                              (768, "lowrank", 0), (5, "constant", 0), (124, "lowrank", 0), (40, "heavytail", 0),
                                              (512, "lowrank", 0), (1024, "lowrank", 1)])
def test_filter_changes_nothing_and_rejects(eng, oracle, monkeypatch, dim, kind_, metric):
    ida, kind = eng
    if kind == "emu" and dim in (768, 1024):
        pytest.skip("768-d / 1024-d under the emulator: covered on the GPU (512-d runs the same fat filtered walk here)")
    pc.use_test_build(monkeypatch)                     # (the walk knobs and the counters exist in the test build only)
    rng = np.random.default_rng(1000 + dim)
    n, ef = S(kind, 260, 20000), S(kind, 12, 100)
    pts = _data(rng, kind_, n, dim)
    q = _data(rng, kind_, S(kind, 12, 600), dim)
    q[0] = pts[n // 2]
    q[1] = pts[1] * np.float32(3.0) + np.float32(11.0)                   # a query far outside the lattice: large |q - q^|, nothing rejected wrongly
    cfg = oracle.default_config(metric=metric, ef_search=ef, ef_construction=S(kind, 16, 100))
    oix = oracle.Index.build(pts, cfg, threads=S(kind, 1, 8))
    want = oix.search(q, threads=S(kind, 1, 8))
    h = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, ida.Builder().metric(metric).ef_search(ef))
    total = [0, 0]
    for name, env in WALKS:
        for flt in ("1", "0"):
            with pc.search_variant(env), monkeypatch.context() as m:
                m.setenv("IDIST_FILTER", flt)
                srch = ida.Search()
                got = h.search_batch(q, srch, counters=True)
                seen, rejected = srch.filter_counts()
            pc.check_search_result(got, want)
            if flt == "0":
                assert seen == 0, name
            else:
                total[0] += seen
                total[1] += rejected
                assert seen <= int(want.counters[:, 0].sum()) and rejected <= seen
    filtered = dim <= 1656                              # thin waves to 496-d, one fat filtered wave per SIMD beyond (768-d, 512-d, 1024-d)
    if not filtered:
        assert total[0] == 0
    elif kind_ == "constant":
        assert total[1] == 0                            # every distance is 0: nothing can be rejected
    else:
        assert total[0] > 0
        if kind_ in ("lowrank", "uniform", "offset"):
            assert total[1] > 0.3 * total[0], total     # the filter is at work (the larger GPU cases reject ~90 %)


def test_filter_unfiltered_geometry_runs_without(eng, oracle, monkeypatch):
    """1700-d rows: no compile-time instantiation and more than thirteen chunks per compact row — the walk runs unfiltered."""
    ida, kind = eng
    pc.use_test_build(monkeypatch)
    rng = np.random.default_rng(5)
    n, dim, ef = S(kind, 100, 3000), 1700, S(kind, 10, 100)
    pts = pc.gen_points(rng, n, dim, "lowrank")
    q = pc.gen_points(rng, S(kind, 6, 200), dim, "lowrank")
    oix = oracle.Index.build(pts, oracle.default_config(ef_search=ef, ef_construction=S(kind, 12, 100)), threads=S(kind, 1, 8))
    h = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, ida.Builder().ef_search(ef))
    with pc.search_variant(WALKS[0][1]):
        srch = ida.Search()
        got = h.search_batch(q, srch, counters=True)
        assert srch.filter_counts() == (0, 0)
    pc.check_search_result(got, oix.search(q))


@pytest.mark.parametrize("dim,kind_,metric", [(300, "lowrank", 0), (128, "uniform", 1), (16, "uniform", 0), (48, "outliers", 0),
                                              (33, "offset", 1), (7, "grid", 0), (768, "lowrank", 0), (100, "nonfinite", 0),
                                              (640, "lowrank", 0)])
def test_filter_bound_never_exceeds_the_canonical_distance(eng, oracle, dim, kind_, metric):
    """idist_filter_bound_batch: what the walk's filter compares with nearest[ef-1], for arbitrary (query, point) pairs.  The bound
    must never exceed the canonical distance of idist_distance_batch (= FloatArray::distance, py/lib.rs:378-421; the oracle's on a
    sample) — for queries from the data's distribution, far outside the lattice, and with non-finite coordinates — and on ordinary
    data it must be tight enough to be of use."""
    ida, kind = eng
    if kind == "emu" and dim == 768:
        pytest.skip("768-d under the emulator: covered on the GPU")
    rng = np.random.default_rng(77 + dim)
    n, nq, n_ids = S(kind, 300, 20000), S(kind, 6, 64), S(kind, 70, 2000)
    base = "lowrank" if kind_ == "nonfinite" else kind_
    pts = _data(rng, base, n, dim)
    q = _data(rng, base, nq, dim)
    q[1] = q[1] * np.float32(5.0) - np.float32(3.0)                       # far outside the lattice
    if kind_ == "nonfinite":
        pts[rng.choice(n, 6, replace=False), rng.integers(0, dim, 6)] = [np.nan, np.inf, -np.inf, np.nan, np.inf, -np.inf]
        q[2, 3], q[3, 0] = np.nan, np.inf
    zero = np.full((n, 64), pc.INVALID, dtype=np.uint32)
    h = ida.Hnsw.from_parts(pts, zero, [], ida.Builder().metric(metric))    # an empty graph is enough for both gather kernels
    ids = rng.integers(0, n, size=(nq, n_ids)).astype(np.uint32)
    ids[0, 1] = pc.INVALID
    ids[2, :4] = [0, 1, 2, 3]
    bound = h.filter_bounds(q, ids)
    dist = h.distances(q, ids)
    d0 = np.array([oracle.distance(q[0], pts[j], metric) if j != pc.INVALID else np.inf for j in ids[0]], dtype=np.float32)
    assert np.array_equal(pc.bits(dist[0]), pc.bits(d0))
    assert np.all(np.isfinite(bound)) and np.all(bound >= 0)
    ok = ~np.isnan(dist)                                                    # NaN distances sort last: any bound is below them
    assert np.all(bound[ok] <= dist[ok]), float(np.max(bound[ok] - dist[ok]))
    assert bound[0, 1] == 0
    if kind_ in ("lowrank", "uniform", "offset"):
        body = bound[4:] / np.maximum(dist[4:], np.float32(1e-30))          # queries from the data's own distribution
        assert np.median(body) > (0.5 if dim < 64 else 0.8), float(np.median(body))
