"""Parity cases shared by the CPU (emulated-kernel) and GPU (-m gpu) test modules.

Every case calls the product through the C ABI (include/idist.h, via the ctypes host
layer) and checks it against the CPU oracle on the same seeded inputs.  Bit-exact for
ids/order/counts/counters; distances compared as raw f32 bits.
"""
import contextlib
import os

import numpy as np

INVALID = 0xFFFFFFFF

# Variants of the graph walk (idist_device.hpp): the default keeps the visited set on chip (LDS hash set of 16-bit
# quotients, single ids overflow to the HBM bitmap — forced early with a tiny set; IDIST_TAB_FORMAT=ids: full ids,
# frozen at 7/8); IDIST_VISITED=bitmap selects the bitmap + Bloom-filter walks (classic / latency / overlap by batch
# width and IDIST_WALK).  All must give the reference's results.
# Wide on-chip walks on the quotient set run with the reject filter in front of their distance passes, as two thin waves per SIMD
# (tests/test_filter.py; IDIST_FILTER=0: the round-5 kernels without it).  The id-set, classic and long-walk variants are unfiltered.
SEARCH_VARIANTS = (("default (narrow batches: four waves per query; wide ones by index size and ef_search)", {}),
                   ("on-chip, quotient set (filtered: two thin waves per SIMD)", {"IDIST_QUAD_NQ": "0", "IDIST_VISITED": "onchip", "IDIST_TAB_FORMAT": "q16"}),
                   ("on-chip, quotient set, no filter", {"IDIST_QUAD_NQ": "0", "IDIST_VISITED": "onchip", "IDIST_TAB_FORMAT": "q16", "IDIST_FILTER": "0"}),
                   ("four waves per query, quotient set", {"IDIST_QUAD_NQ": "4000000000", "IDIST_TAB_FORMAT": "q16"}),
                   ("long-walk form: two 256-register waves per SIMD on the quotient set (policy: ef_search >= 512 at 300-d, unfiltered indexes)",
                    {"IDIST_QUAD_NQ": "0", "IDIST_VISITED": "onchip", "IDIST_TAB_FORMAT": "q16", "IDIST_W2_EF": "0", "IDIST_FILTER": "0"}),
                   ("four waves per query, 512-B quotient set (256 ids) then bitmap", {"IDIST_QUAD_NQ": "4000000000", "IDIST_TAB_LOG2": "7"}),
                   ("on-chip classic", {"IDIST_WALK": "classic", "IDIST_VISITED": "onchip"}),
                   ("on-chip, full ids", {"IDIST_QUAD_NQ": "0", "IDIST_VISITED": "onchip", "IDIST_TAB_FORMAT": "ids"}),
                   ("four waves per query, full ids, set of 128 ids then bitmap",
                    {"IDIST_QUAD_NQ": "4000000000", "IDIST_TAB_LOG2": "7", "IDIST_TAB_FORMAT": "ids"}),
                   ("on-chip classic, full ids, set of 32 ids", {"IDIST_TAB_LOG2": "5", "IDIST_WALK": "classic", "IDIST_TAB_FORMAT": "ids"}),
                   ("on-chip, 512-B quotient set (256 ids) then bitmap", {"IDIST_TAB_LOG2": "7", "IDIST_QUAD_NQ": "0"}),
                   ("on-chip classic, 128-B quotient set (64 ids): bitmap almost from the start", {"IDIST_TAB_LOG2": "5", "IDIST_WALK": "classic"}),
                   ("bitmap overlap", {"IDIST_VISITED": "bitmap", "IDIST_LATENCY_NQ": "0"}),
                   ("bitmap latency", {"IDIST_VISITED": "bitmap", "IDIST_LATENCY_NQ": "4000000000"}),
                   ("bitmap classic", {"IDIST_VISITED": "bitmap", "IDIST_LATENCY_NQ": "0", "IDIST_WALK": "classic"}))
# The descent of an insertion shares the walk code; what varies there is the walk mode (classic / overlap) and the size
# of the on-chip visited set — it never runs four waves per item and has no bitmap-only variant.
# (the third field: the variant runs every selection in the reference's own order, so its count of distance calls inside
#  select_heuristic / add_neighbor_heuristic must equal the oracle's n_heur)
BUILD_VARIANTS = (("on-chip (narrow steps: four waves per insertion)", {}),
                  ("on-chip, one wave per insertion also in narrow steps (descents with the reject filter where the policy has it: rows >= 192 floats)", {"IDIST_BUILD_QUAD": "0"}),
                  ("on-chip, one wave per insertion, descents WITH the reject filter at every row length", {"IDIST_BUILD_QUAD": "0", "IDIST_BUILD_FILTER": "1"}),
                  ("on-chip, one wave per insertion, descents WITHOUT the reject filter", {"IDIST_BUILD_QUAD": "0", "IDIST_BUILD_FILTER": "0"}),
                  ("on-chip, 512-register descent waves", {"IDIST_BUILD_QUAD": "0", "IDIST_BUILD_A_REGS": "512"}),
                  ("reference-order kernels: LDS-tile selection, every update from scratch", {"IDIST_BUILD_A2": "tile", "IDIST_BUILD_NO_FAST": "1"}),
                  ("step A2 with the LDS-tile kernel instead of the Gram matrix on MFMA", {"IDIST_BUILD_A2": "tile"}),
                  ("on-chip classic", {"IDIST_WALK": "classic"}),
                  ("on-chip, full ids (frozen at 7/8, distance log in the id form)", {"IDIST_TAB_FORMAT": "ids"}),
                  ("on-chip, full ids, small set then bitmap", {"IDIST_TAB_FORMAT": "ids", "IDIST_TAB_LOG2": "7"}),
                  ("on-chip, smallest set the build allows (IDIST_TAB_LOG2 is clamped to 8: 512 quotients) then bitmap", {"IDIST_TAB_LOG2": "7"}),
                  ("on-chip classic, smallest set then bitmap", {"IDIST_TAB_LOG2": "5", "IDIST_WALK": "classic"}))


def _emulated():
    """True when the engine under test is the CPU lockstep emulator (tests/simt): a build there costs seconds per variant."""
    from instant_distance_amd import _capi
    return _capi.lib().path.endswith("libidist_emu.so")


def pick_variants(variants, seed, keep):
    """GPU: every variant for every case.  Emulator: the default plus `keep` others, rotating with the case's seed, so that
    the CPU suite stays within minutes while every variant is still run by several cases."""
    vs = list(variants)
    if not _emulated() or len(vs) <= keep + 1:
        return vs
    rest = vs[1:]
    picked = []
    for i in range(len(rest)):
        v = rest[(seed * 5 + i * 7) % len(rest)]           # 7 is coprime to both list lengths
        if v not in picked:
            picked.append(v)
        if len(picked) == keep:
            break
    return [vs[0]] + picked


_VARIANTS_LIB = None


def variants_lib():
    """libidist_variants.so: the product's sources + the walks no policy selects (`make variants`, built by
    __graft_entry__.build()).  Test infrastructure; the product package never loads it."""
    global _VARIANTS_LIB
    if _VARIANTS_LIB is None:
        from instant_distance_amd import _capi
        path = os.path.join(os.path.dirname(_capi.LIB_PATH), "libidist_variants.so")
        assert os.path.exists(path), "libidist_variants.so is missing: run __graft_entry__.build() (make -C instant-distance_amd/csrc variants)"
        _VARIANTS_LIB = _capi.Lib(path)
    return _VARIANTS_LIB


def use_test_build(monkeypatch):
    """For the rest of this test the engine is the TEST build: libidist.so reads no IDIST_* knob but IDIST_COMBINE / IDIST_SYNC /
    IDIST_KERNEL_EVENTS (the others are compiled out of the product), so a GPU test that steers kernels or schedules through the
    environment runs libidist_variants.so — the same sources and struct layouts plus the knobs and the walks no policy selects.
    Handles made by either library are valid in the other.  (The emulator build holds every knob and variant.)"""
    if not _emulated():
        from instant_distance_amd import _capi
        _capi.lib()
        monkeypatch.setattr(_capi, "_singleton", variants_lib())


@contextlib.contextmanager
def search_variant(env):
    """Environment of one walk variant.  A non-empty environment selects kernels through test knobs, which exist in the test build
    only: on the GPU the block runs through libidist_variants.so (see use_test_build: an index imported before the block is searched
    inside it, an index built inside it is exported after it); the default variant ({}) runs the product library."""
    if isinstance(env, str):                     # a bare IDIST_LATENCY_NQ value
        env = {"IDIST_LATENCY_NQ": env}
    from instant_distance_amd import _capi
    swapped = None
    if env and not _emulated():
        swapped = _capi.lib()                    # (loads the product library if nothing is loaded yet: there is always one to restore)
        if swapped is not variants_lib():
            _capi._singleton = variants_lib()
        else:
            swapped = None
    keys = ("IDIST_LATENCY_NQ", "IDIST_WALK", "IDIST_VISITED", "IDIST_TAB_LOG2", "IDIST_QUAD_NQ", "IDIST_BUILD_A2",
            "IDIST_TAB_FORMAT", "IDIST_BUILD_NO_FAST", "IDIST_BUILD_QUAD", "IDIST_BUILD_A_REGS", "IDIST_W2_EF", "IDIST_FILTER",
            "IDIST_BUILD_FILTER")
    old = {k: os.environ.get(k) for k in keys}
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        yield
    finally:
        if swapped is not None:
            _capi._singleton = swapped
        for k in keys:
            os.environ.pop(k, None)
            if old[k] is not None:
                os.environ[k] = old[k]


def check_search_result(got, want):
    assert np.array_equal(got.count, want.count)
    assert np.array_equal(got.pid, want.pid)
    assert np.array_equal(bits(got.distance), bits(want.dist))
    assert np.array_equal(got.counters, want.counters)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def gen_points(rng, n, dim, kind="uniform"):
    if kind == "uniform":        # what the reference's tests use (tests/all.rs:59, test.py:5)
        return rng.random((n, dim), dtype=np.float32)
    if kind == "grid":           # integer coordinates: exact ties in the distance (examples/colors.rs style)
        return rng.integers(0, 6, size=(n, dim)).astype(np.float32)
    if kind == "lowrank":        # fastText-shape: low intrinsic dimension, normalised rows
        z = rng.standard_normal((n, 8)).astype(np.float32)
        a = rng.standard_normal((8, dim)).astype(np.float32)
        x = z @ a + 0.05 * rng.standard_normal((n, dim)).astype(np.float32)
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    raise ValueError(kind)


def check_distance_batch(ida, oracle, n, dim, metric, seed=0, nq=5, n_ids=70):
    rng = np.random.default_rng(seed)
    pts = gen_points(rng, n, dim)
    b = ida.Builder().metric(metric).max_batch(1)
    # an index with an empty graph is enough for the gather-L2 kernel
    zero = np.full((n, 64), INVALID, dtype=np.uint32)
    h = ida.Hnsw.from_parts(pts, zero, [], b)
    q = gen_points(rng, nq, dim)
    ids = rng.integers(0, n, size=(nq, n_ids)).astype(np.uint32)
    ids[0, 3] = INVALID
    ids[-1, -1] = INVALID
    got = h.distances(q, ids)
    want = np.array([[oracle.distance(q[i], pts[j], metric) if j != INVALID else np.inf for j in ids[i]]
                     for i in range(nq)], dtype=np.float32)
    assert np.array_equal(bits(got), bits(want))


_ORACLE_CACHE = {}


def oracle_graph(oracle, n, dim, kind, metric, seed, threads, ef_construction):
    """Oracle-built graph, cached per parameter set (the oracle build dominates test time)."""
    key = (n, dim, kind, metric, seed, threads, ef_construction)
    if key not in _ORACLE_CACHE:
        rng = np.random.default_rng(seed)
        pts = gen_points(rng, n, dim, kind)
        cfg = oracle.default_config(metric=metric, ef_construction=ef_construction)
        _ORACLE_CACHE.clear()            # keep at most one graph alive
        _ORACLE_CACHE[key] = (pts, oracle.Index.build(pts, cfg, threads=threads), rng)
    return _ORACLE_CACHE[key]


def check_search_parity(ida, oracle, n, dim, ef_search=100, metric=0, kind="uniform", nq=16, seed=0, threads=1,
                        ef_construction=100, graph_seed=None):
    """Oracle builds the graph; engine imports it; same queries must give identical results."""
    pts, oix, _ = oracle_graph(oracle, n, dim, kind, metric, seed if graph_seed is None else graph_seed, threads,
                               ef_construction)
    oix.set_ef_search(ef_search)
    rng = np.random.default_rng(seed + 1000003)
    b = ida.Builder().metric(metric).ef_search(ef_search)
    h = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, b)
    q = gen_points(rng, nq, dim, kind)
    if kind == "uniform" and nq > 2:
        q[1] = pts[min(5, n - 1)]          # a stored point: distance 0 first (test.py:15-35)
    want = oix.search(q, threads=1)
    for _, lat in pick_variants(SEARCH_VARIANTS, seed, 7):
        with search_variant(lat):
            got = h.search_batch(q, ida.Search(), counters=True)
        check_search_result(got, want)
    # sortedness + idempotence (size independent properties)
    for i in range(nq):
        c = int(got.count[i])
        d = got.distance[i, :c]
        assert np.all(d[:-1] <= d[1:])
        assert len(set(got.pid[i, :c].tolist())) == c
    again = h.search_batch(q, ida.Search())
    assert np.array_equal(again.pid, got.pid)
    return h, oix


def check_build_exact(ida, oracle, n, dim, metric=0, kind="uniform", ef_construction=100, keep_pruned=True, seed=0,
                      heuristic=True, extend=False, max_batch=1, variants=None):
    """max_batch = 1: zero/layers byte-identical to the oracle's sequential build."""
    rng = np.random.default_rng(seed)
    pts = gen_points(rng, n, dim, kind)
    cfg = oracle.default_config(metric=metric, ef_construction=ef_construction, keep_pruned=int(keep_pruned),
                                has_heuristic=int(heuristic), extend_candidates=int(extend))
    oix = oracle.Index.build(pts, cfg, threads=1)
    b = (ida.Builder().metric(metric).max_batch(max_batch).ef_construction(ef_construction)
         .select_heuristic(ida.Heuristic(extend, keep_pruned) if heuristic else None))
    for _, lat in (variants or pick_variants(BUILD_VARIANTS, seed, 4)):   # the descent of an insertion has the same variants as the search
        with search_variant(lat):
            h = ida.Hnsw.from_ordered_points(pts, b)
        zero, layers = h.into_parts()
        assert np.array_equal(zero, oix.zero)
        assert len(layers) == len(oix.layers)
        for a, o in zip(layers, oix.layers):
            assert np.array_equal(a, o)
        st = h.build_stats()
        assert st.n_dist == oix.build_counters.n_dist
        assert st.n_exp0 == oix.build_counters.n_exp0 and st.n_expU == oix.build_counters.n_expU
        # the reference's own count of distance calls inside select_heuristic + add_neighbor_heuristic (early-exit `any`,
        # core/lib.rs:676-679): defined where every selection ran in the reference's order
        if heuristic and (extend or (max_batch == 1 and "IDIST_BUILD_NO_FAST" in lat and lat.get("IDIST_BUILD_A2") == "tile")):
            assert st.n_heur_ref == oix.build_counters.n_heur, (st.n_heur_ref, oix.build_counters.n_heur)
        elif not extend:
            assert st.n_heur_ref == 0
    return h


def recall_at(found_pid, truth_pid, k):
    hit = 0
    for f, t in zip(found_pid, truth_pid):
        hit += len(set(f[:k].tolist()) & set(t[:k].tolist()))
    return hit / (k * len(truth_pid))


def check_build_batched(ida, oracle, n, dim, max_batch=0, kind="uniform", k=10, nq=50, seed=0, min_recall=0.95):
    """Concurrent inserts (the rayon path, core/lib.rs:316-318): judged by recall + degree."""
    rng = np.random.default_rng(seed)
    pts = gen_points(rng, n, dim, kind)
    q = gen_points(rng, nq, dim, kind)
    h = ida.Hnsw.from_ordered_points(pts, ida.Builder().max_batch(max_batch))
    truth, _ = h.bruteforce(q, k)
    o_truth, _ = oracle.bruteforce(pts, q, k)
    assert np.array_equal(truth, o_truth)
    got = h.search_batch(q, ida.Search())
    rec = recall_at(got.pid, truth, k)
    assert rec >= min_recall, rec
    zero, layers = h.into_parts()
    # graph invariants of the reference builder
    assert zero.shape == (n, 64)
    valid = zero != INVALID
    assert np.all(valid[:, :-1] >= valid[:, 1:])                       # prefix-valid rows
    assert np.all(zero[valid] < n)
    for r in range(0, n, max(1, n // 50)):
        ids = zero[r][valid[r]]
        assert len(set(ids.tolist())) == len(ids) and r not in ids    # unique, no self loop
    sizes = oracle.layer_sizes(n)
    assert [l.shape[0] for l in layers] == sizes[1:]
    # deterministic given the batch schedule
    h2 = ida.Hnsw.from_ordered_points(pts, ida.Builder().max_batch(max_batch))
    z2, l2 = h2.into_parts()
    assert np.array_equal(zero, z2)
    return rec


def check_build_concurrent_invariants(ida, oracle, n, dim, kind="uniform", metric=0, ef_construction=100, keep_pruned=True,
                                      max_batch=0, seed=0, nq=200, k=10, slack=0.03):
    """Default (concurrent, pipelined) build: no oracle graph to compare bytes with, so check what must hold for ANY
    legal outcome — the reference's row invariants (re-import validates: ids in range, no duplicates before the first
    INVALID), determinism, and recall@k no worse than the oracle's threaded build of the same points (minus slack)."""
    rng = np.random.default_rng(seed)
    pts = gen_points(rng, n, dim, kind)
    q = gen_points(rng, nq, dim, kind)
    b = (ida.Builder().metric(metric).max_batch(max_batch).ef_construction(ef_construction)
         .select_heuristic(ida.Heuristic(False, keep_pruned)))
    h = ida.Hnsw.from_ordered_points(pts, b)
    zero, layers = h.into_parts()
    zero2, layers2 = ida.Hnsw.from_ordered_points(pts, b).into_parts()
    assert np.array_equal(zero, zero2) and all(np.array_equal(x, y) for x, y in zip(layers, layers2))
    ida.Hnsw.from_parts(pts, zero, layers, b)                       # idist_index_import validates every row
    assert not np.any(zero[:, 0] == INVALID) or n <= 1              # every point has a neighbour
    own = np.arange(n, dtype=np.uint32)[:, None]
    assert not np.any(zero == own)                                  # no self links
    truth, _ = h.bruteforce(q, k)
    got = h.search_batch(q, ida.Search())
    rec = recall_at(got.pid, truth, k)
    cfg = oracle.default_config(metric=metric, ef_construction=ef_construction, keep_pruned=int(keep_pruned))
    oix = oracle.Index.build(pts, cfg, threads=4)
    orec = recall_at(oix.search(q).pid, truth, k)
    assert rec >= orec - slack, (rec, orec)
    return rec, orec
