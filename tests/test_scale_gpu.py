"""Parity at sizes the small cases cannot reach (`-m gpu` only).

  * `Hnsw::new` / `Construction::insert` (core/lib.rs:304-329, 437-528) in exact mode (`max_batch = 1`) at 100,000 x 300 and
    40,000 x 768: six / five layers, descents that fill the on-chip visited set, the distance log at scale.  zero and all
    upper layers must be byte-identical to the oracle's sequential build (= the reference with one rayon thread), and
    the work counters {n_dist, n_exp0, n_expU} equal.
  * `Hnsw::search` (core/lib.rs:352-383) at dimensions NO compile-time geometry exists for (384, 1024: the runtime-geometry
    walk) and real size: the oracle searches the exported graph; ids, order, counts, distance bits and work counters
    must be identical at two ef_search values (id set / quotient set).

The oracle's exact builds take minutes even with `threads = -8` (insertions strictly sequential, the independent neighbour
updates of one insertion on 8 threads: the same bytes as `threads = 1`, pinned by tests/test_oracle_golden.py): they run
in background threads (ctypes drops the GIL) from the moment the module's first test starts, while the GPU inserts the
same points."""
import concurrent.futures as cf

import numpy as np
import pytest

import parity_cases as pc

EXACT_CASES = [(100_000, 300), (40_000, 768)]


def fasttext_shape(n, dim, seed):
    """bench.py's synthetic rows: 32-d latent + 5 % noise, L2-normalised."""
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((n, 32), dtype=np.float32)
    a = np.random.default_rng(4242).standard_normal((32, dim), dtype=np.float32)
    pts = z @ a
    pts += np.float32(0.05) * rng.standard_normal((n, dim), dtype=np.float32)
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    return np.ascontiguousarray(pts, dtype=np.float32)


@pytest.fixture(scope="module")
def oracle_builds(oracle):
    pool = cf.ThreadPoolExecutor(len(EXACT_CASES))
    jobs = {}
    for n, dim in EXACT_CASES:
        pts = fasttext_shape(n, dim, 7 + dim)
        jobs[(n, dim)] = (pts, pool.submit(oracle.Index.build, pts, oracle.default_config(), -8))
    yield jobs
    pool.shutdown(wait=True, cancel_futures=True)


@pytest.mark.gpu
@pytest.mark.parametrize("n,dim", EXACT_CASES)
def test_build_exact_gpu_large(engine_loader, oracle, oracle_builds, n, dim):
    ida = engine_loader("gpu")
    pts, job = oracle_builds[(n, dim)]
    h = ida.Hnsw.from_ordered_points(pts, ida.Builder().max_batch(1))
    zero, layers = h.into_parts()
    st = h.build_stats()
    oix = job.result()
    assert len(layers) == len(oix.layers) == len(oracle.layer_sizes(n)) - 1
    assert np.array_equal(zero, oix.zero), f"{int((zero != oix.zero).any(1).sum())} of {n} zero rows differ"
    for l, (a, o) in enumerate(zip(layers, oix.layers)):
        assert np.array_equal(a, o), f"layer {l + 1} differs"
    assert (st.n_dist, st.n_exp0, st.n_expU) == (oix.build_counters.n_dist, oix.build_counters.n_exp0, oix.build_counters.n_expU)
    assert st.n_batches == n - 1 and st.tie_overflow == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,dim,nq", [(200_000, 384, 2000), (60_000, 1024, 1000), (50_000, 100, 1000)])
def test_search_parity_runtime_geometry_at_size_gpu(engine_loader, oracle, n, dim, nq):
    ida = engine_loader("gpu")
    pts = fasttext_shape(n, dim, 11)
    q = fasttext_shape(nq, dim, 12)
    q[:32] = pts[:32]
    h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
    zero, layers = h.into_parts()
    assert [l.shape[0] for l in layers] == oracle.layer_sizes(n)[1:]
    oix = oracle.Index.from_arrays(pts, zero, layers, oracle.default_config())
    for ef in (100, 250):
        h.set_ef_search(ef)
        oix.set_ef_search(ef)
        want = oix.search(q, threads=8)
        got = h.search_batch(q, ida.Search(), counters=True)                 # wide batch: one wave per query
        pc.check_search_result(got, want)
        narrow = h.search_batch(q[:64], ida.Search(), counters=True)         # narrow batch: four waves per query
        assert np.array_equal(narrow.pid, want.pid[:64]) and np.array_equal(narrow.counters, want.counters[:64])
        assert np.array_equal(pc.bits(narrow.distance), pc.bits(want.dist[:64]))
        assert np.all(got.count == ef) and np.all(got.distance[:, :-1] <= got.distance[:, 1:])
        assert np.array_equal(got.pid[:32, 0], np.arange(32)) and np.all(got.distance[:32, 0] == 0)   # stored points find themselves
    truth, _ = h.bruteforce(q[:512], 10)
    assert pc.recall_at(got.pid[:512], truth, 10) > 0.9


@pytest.mark.gpu
@pytest.mark.parametrize("n,dim,efs", [(200_000, 300, (600, 800, 1000, 1024, 513)), (120_000, 384, (600, 800))])
def test_search_parity_long_walks_ef_600_to_1000_gpu(engine_loader, oracle, n, dim, efs):
    """ef_search in (512, 1024] in steady state, default policy (no environment knob): the sixteen-register-block merge of
    `Search::push` (core/lib.rs:704-720) and the two-256-register-waves-per-SIMD layout the policy picks from ef 512 at 300-d.
    200,000 x 300 fastText-shape (and 120,000 x 384: the runtime-geometry rows take the same layout from ef 512 on), GPU build, the
    oracle searches the exported graph (core/lib.rs:598-614); wide batch (one wave per query) and a 64-query narrow batch (four
    waves per query): ids, order, counts, distance bits, work counters."""
    import os

    for knob in ("IDIST_W2_EF", "IDIST_LATENCY_NQ", "IDIST_QUAD_NQ", "IDIST_VISITED", "IDIST_TAB_FORMAT", "IDIST_TAB_LOG2"):
        assert knob not in os.environ, f"{knob} is set: this case pins the DEFAULT policy"
    ida = engine_loader("gpu")
    nq = 2048
    pts = fasttext_shape(n, dim, 21)
    q = fasttext_shape(nq, dim, 22)
    h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
    zero, layers = h.into_parts()
    oix = oracle.Index.from_arrays(pts, zero, layers, oracle.default_config())
    for ef in efs:
        h.set_ef_search(ef)
        oix.set_ef_search(ef)
        want = oix.search(q, threads=16)
        got = h.search_batch(q, ida.Search(), counters=True)                 # wide batch
        pc.check_search_result(got, want)
        assert np.all(got.count == ef) and np.all(got.distance[:, :-1] <= got.distance[:, 1:])
        narrow = h.search_batch(q[:64], ida.Search(), counters=True)         # narrow batch
        assert np.array_equal(narrow.pid, want.pid[:64]) and np.array_equal(narrow.counters, want.counters[:64])
        assert np.array_equal(pc.bits(narrow.distance), pc.bits(want.dist[:64])) and np.array_equal(narrow.count, want.count[:64])
        # device-pointer style reuse: the same Search answers a second ef-wide batch identically (steady state, not first use)
        s = ida.Search()
        a = h.search_batch(q[:1500], s, counters=True)
        b = h.search_batch(q[:1500], s, counters=True)
        assert np.array_equal(a.pid, b.pid) and np.array_equal(a.pid, want.pid[:1500]) and np.array_equal(b.counters, want.counters[:1500])


def _growth_data(kind, n, dim, seed):
    rng = np.random.default_rng(seed)
    if kind == "clustered":                      # 40 tight clusters: a new point's true neighbours are mostly points of its own cluster
        centres = rng.standard_normal((40, dim)).astype(np.float32) * 4
        return (centres[rng.integers(0, 40, n)] + 0.1 * rng.standard_normal((n, dim)).astype(np.float32)).astype(np.float32)
    if kind == "lowrank":
        return pc.gen_points(rng, n, dim, "lowrank")
    if kind == "duplicates":                     # every point exists four times (exact ties at distance 0) plus jittered copies
        base = rng.random((n // 8, dim), dtype=np.float32)
        reps = np.concatenate([base] * 4 + [base + 1e-3 * rng.standard_normal(base.shape).astype(np.float32)] * 4)
        return np.ascontiguousarray(reps[rng.permutation(len(reps))][:n], dtype=np.float32)
    raise ValueError(kind)


def _reachable_from_entry(zero):
    n = zero.shape[0]
    seen = np.zeros(n, dtype=bool)
    seen[0] = True
    frontier = np.array([0], dtype=np.int64)
    while frontier.size:
        nxt = zero[frontier].reshape(-1)
        nxt = nxt[nxt != 0xFFFFFFFF].astype(np.int64)
        nxt = np.unique(nxt[~seen[nxt]])
        seen[nxt] = True
        frontier = nxt
    return int(seen.sum())


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["clustered", "lowrank", "duplicates"])
@pytest.mark.parametrize("n", [2000, 20000])
def test_growth_phase_quality_on_hard_data_gpu(engine_loader, oracle, monkeypatch, kind, n):
    """Narrow steps hold g/8 insertions (lag 2: a descent can miss the newest quarter of the graph while it is small), wide ones
    g/32.  At these sizes the growth phase IS the build, and uniform data is the easy case: clustered, low-rank and
    duplicate-heavy points, default schedule (product library) against the old g/32 rule (test build, IDIST_BUILD_GROWTH=32) and
    the CPU oracle's threaded build (the reference's rayon path, core/lib.rs:316-318) — recall@10 of the same queries through the
    same engine, and every point reachable from the entry point on the zero layer."""
    ida = engine_loader("gpu")
    dim = 32
    pts = _growth_data(kind, n, dim, 5)
    q = _growth_data(kind, 1000, dim, 6)
    h8 = ida.Hnsw.from_ordered_points(pts, ida.Builder())
    truth, _ = h8.bruteforce(q, 10)
    recall = {"g/8": pc.recall_at(h8.search_batch(q, ida.Search()).pid, truth, 10)}
    reach = {"g/8": _reachable_from_entry(h8.into_parts()[0])}
    oix = oracle.Index.build(pts, oracle.default_config(), threads=8)
    ho = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, ida.Builder())
    recall["cpu threaded"] = pc.recall_at(ho.search_batch(q, ida.Search()).pid, truth, 10)
    reach["cpu threaded"] = _reachable_from_entry(oix.zero)
    pc.use_test_build(monkeypatch)
    monkeypatch.setenv("IDIST_BUILD_GROWTH", "32")
    h32 = ida.Hnsw.from_ordered_points(pts, ida.Builder())
    recall["g/32"] = pc.recall_at(h32.search_batch(q, ida.Search()).pid, truth, 10)
    reach["g/32"] = _reachable_from_entry(h32.into_parts()[0])
    assert h32.build_stats().n_batches > h8.build_stats().n_batches          # (the knob took effect: more, smaller steps)
    assert recall["g/8"] >= recall["g/32"] - 0.01 and recall["g/8"] >= recall["cpu threaded"] - 0.01, (recall, reach)
    # select_heuristic gives no connectivity guarantee (a point every neighbour prunes has no in-link: the reference's graphs have
    # such points too on clustered data), so reachability is held against the CPU build's, not against n
    assert reach["g/8"] >= min(reach["cpu threaded"], reach["g/32"]) - n // 200, (reach, recall)


@pytest.mark.gpu
def test_34m_points_beyond_the_quotient_sets_reach_gpu(engine_loader, oracle):
    """n = 34,000,000 x 4-d: beyond what the 16-bit quotient form of the on-chip visited set can address (33.5 M with the 32-KB
    set: id-form fallback in the search AND in every build descent), 512-B dirty blocks in the overflow bitmap, point and
    adjacency row offsets past 2^32 bytes (zero layer 8.7 GB).  The reference's only bound is n < u32::MAX (core/lib.rs:256).
    Default build on the GPU; the oracle searches the EXPORTED graph (points and zero layer borrowed in place) for 256 queries at
    ef_search 100 and 300: ids, order, counts, distance bits and work counters must be identical, wide and narrow batches; plus the
    size-independent properties of the graph on a strided sample of rows."""
    import time

    ida = engine_loader("gpu")
    n, dim, nq = 34_000_000, 4, 256
    rng = np.random.default_rng(34)
    pts = rng.random((n, dim), dtype=np.float32)
    q = rng.random((nq, dim), dtype=np.float32)
    q[:32] = pts[np.linspace(0, n - 1, 32).astype(np.int64)]               # stored points from the whole id range
    t0 = time.time()
    h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
    st = h.build_stats()
    print(f"34M x 4-d: built in {st.seconds:.1f} s on the device ({time.time() - t0:.1f} s with the upload), {st.n_batches} steps")
    zero, layers = h.into_parts()
    sizes = oracle.layer_sizes(n)
    assert [l.shape[0] for l in layers] == sizes[1:] and zero.shape == (n, 64)
    rows = zero[:: 17]                                                       # 2M rows from the whole range
    valid = rows != pc.INVALID
    assert np.all(valid[:, :-1] >= valid[:, 1:]) and np.all(rows[valid] < n) and np.all(valid[:, 0])
    assert not np.any(rows == (np.arange(0, n, 17, dtype=np.uint32)[:, None]))
    assert np.all(zero[-1000:][zero[-1000:] != pc.INVALID] < n)              # the last rows: offsets beyond 8.7e9 bytes
    oix = oracle.Index.from_arrays(pts, zero, layers, oracle.default_config(), borrow=True)
    for ef in (100, 300):
        h.set_ef_search(ef)
        oix.set_ef_search(ef)
        want = oix.search(q, threads=16)
        s = ida.Search()
        got = h.search_batch(q, s, counters=True)                            # wide batch
        pc.check_search_result(got, want)
        again = h.search_batch(q, s, counters=True)                          # the same Search again: the visited set was cleared
        pc.check_search_result(again, want)
        narrow = h.search_batch(q[:8], ida.Search(), counters=True)          # four waves per query
        assert np.array_equal(narrow.pid, want.pid[:8]) and np.array_equal(narrow.counters, want.counters[:8])
        assert np.array_equal(pc.bits(narrow.distance), pc.bits(want.dist[:8]))
        assert np.all(got.count == ef) and np.all(got.distance[:, :-1] <= got.distance[:, 1:])
        assert np.all(got.distance[:32, 0] == 0)                             # stored points find themselves (or an exact duplicate)
    truth, _ = h.bruteforce(q[:64], 10)
    assert pc.recall_at(got.pid[:64], truth, 10) > 0.9
