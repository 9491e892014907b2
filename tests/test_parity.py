"""HIP path vs CPU oracle through the C ABI (include/idist.h).

CPU (`-m "not gpu"`): the real kernel source under the lockstep emulator, tiny sizes.
GPU (`-m gpu`): libidist.so on the MI355X at the sizes the oracle finishes in seconds.
Bar: ids / order / counts / work counters bit-exact, distances equal as raw f32 bits.
"""
import numpy as np
import pytest

import parity_cases as pc
from engines import engine_params


@pytest.fixture(params=engine_params())
def eng(request, engine_loader):
    ida = engine_loader(request.param)
    return ida, request.param


def S(kind, emu, gpu):
    return gpu if kind == "gpu" else emu


@pytest.mark.parametrize("dim", [1, 2, 3, 4, 7, 13, 32, 100, 128, 300, 301, 768])
@pytest.mark.parametrize("metric", [0, 1])
def test_distance_batch(eng, oracle, dim, metric):
    ida, kind = eng
    pc.check_distance_batch(ida, oracle, n=S(kind, 40, 3000), dim=dim, metric=metric, seed=dim,
                            nq=S(kind, 3, 9), n_ids=S(kind, 70, 200))


@pytest.mark.parametrize("dim,kind_", [(2, "uniform"), (16, "uniform"), (128, "uniform"), (300, "uniform"),
                                       (300, "lowrank"), (768, "uniform"), (3, "grid"), (2, "grid")])
def test_search_parity(eng, oracle, dim, kind_):
    ida, kind = eng
    n = S(kind, 260, {2: 20000, 3: 20000, 16: 20000, 128: 12000, 300: 8000, 768: 3000}[dim])
    metric = 1 if kind_ == "grid" else 0
    pc.check_search_parity(ida, oracle, n=n, dim=dim, kind=kind_, metric=metric, nq=S(kind, 6, 400), seed=dim)


@pytest.mark.parametrize("ef", [1, 2, 10, 64, 100, 150, 500])
def test_search_parity_ef_sweep(eng, oracle, ef):
    ida, kind = eng
    pc.check_search_parity(ida, oracle, n=S(kind, 300, 20000), dim=S(kind, 8, 128), ef_search=ef,
                           nq=S(kind, 5, 300), seed=ef, graph_seed=77)


def test_search_parity_ef_beyond_merge_width(eng, oracle):
    # W longer than the one-pass merge of `push` covers (1024 entries in the fat waves, 512 elsewhere): on the GPU ef 1500 takes the
    # sequential insertion path in every variant, under the emulator ef 600 does where the merge is 512 wide
    ida, kind = eng
    pc.check_search_parity(ida, oracle, n=S(kind, 640, 30000), dim=S(kind, 4, 64), ef_search=S(kind, 600, 1500),
                           nq=S(kind, 3, 64), seed=41)


def test_search_parity_heavy_ties(eng, oracle):
    # few distinct coordinates => many exactly equal distances: exercises the distance-only
    # break test (core/lib.rs:600-604) and the (distance, pid) tie-break (core/types.rs:229-234)
    ida, kind = eng
    rng = np.random.default_rng(3)
    n = S(kind, 250, 5000)
    pts = rng.integers(0, 3, size=(n, 4)).astype(np.float32)
    cfg = oracle.default_config(metric=1, ef_search=20)
    oix = oracle.Index.build(pts, cfg)
    h = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, ida.Builder().metric(1).ef_search(20))
    q = rng.integers(0, 3, size=(S(kind, 8, 200), 4)).astype(np.float32)
    want = oix.search(q)
    # the default (strict) policy never answers with an error here: a host-pointer batch that overflows the 64-entry tie
    # region is searched again with a larger one (and, past 4096 entries, with the HBM bags) until it IS the reference's
    for _, lat in pc.SEARCH_VARIANTS:
        with pc.search_variant(lat):
            got = h.search_batch(q, ida.Search(), counters=True)
        pc.check_search_result(got, want)
    # with a larger tie region requested up front (idist_config.tie_capacity) the same
    hb = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, ida.Builder().metric(1).ef_search(20).tie_capacity(2048))
    for _, lat in pc.SEARCH_VARIANTS:
        with pc.search_variant(lat):
            pc.check_search_result(hb.search_batch(q, ida.Search(), counters=True), want)


def test_duplicate_points(eng, oracle):
    """30 % exact duplicates: zero distances and (distance, pid) ties everywhere — build (max_batch = 1)
    and search must still match the oracle bit for bit."""
    ida, kind = eng
    rng = np.random.default_rng(21)
    n, dim = S(kind, 160, 4000), S(kind, 5, 24)
    base = rng.random((n, dim), dtype=np.float32)
    dup = rng.integers(0, n, size=n)
    mask = rng.random(n) < 0.3
    pts = np.where(mask[:, None], base[dup], base).astype(np.float32)
    oix = oracle.Index.build(pts, oracle.default_config(ef_search=50))
    h = ida.Hnsw.from_ordered_points(pts, ida.Builder().max_batch(1).ef_search(50))
    zero, layers = h.into_parts()
    assert np.array_equal(zero, oix.zero)
    q = np.concatenate([pts[: S(kind, 6, 100)], rng.random((S(kind, 4, 100), dim), dtype=np.float32)])
    want = oix.search(q)
    for _, lat in pc.SEARCH_VARIANTS:
        with pc.search_variant(lat):
            got = h.search_batch(q, ida.Search(), counters=True)
        pc.check_search_result(got, want)


def test_search_on_parallel_built_graph(eng, oracle):
    # graphs from the rayon-style build (core/lib.rs:316-318) are not distance-sorted per row
    ida, kind = eng
    pc.check_search_parity(ida, oracle, n=S(kind, 300, 20000), dim=S(kind, 6, 64), threads=4, nq=S(kind, 5, 200), seed=9)


@pytest.mark.parametrize("n,dim,kw", [
    (1, 4, {}), (2, 4, {}), (5, 2, {"metric": 1}), (33, 3, {}), (100, 2, {"metric": 1}),
    (150, 12, {}), (120, 300, {}), (140, 8, {"ef_construction": 20}), (130, 5, {"keep_pruned": False}),
    # 700-d: a runtime geometry beyond the thin filter tile (fat, unfiltered descents; the search takes the fat filtered walk)
    (110, 700, {"kind": "lowrank"}),
    (140, 3, {"kind": "grid", "metric": 1}),
    # squared-L2 grids: exact ties d(c_i, c_j) == d(c_i, q) all over — the MFMA selection filter must send them to the
    # canonical distance and keep the strict `<` of core/lib.rs:678
    (140, 3, {"kind": "grid"}), (110, 2, {"kind": "grid", "ef_construction": 40}),
    (150, 6, {"heuristic": False}), (120, 2, {"heuristic": False, "metric": 1}), (100, 300, {"heuristic": False}),
    (130, 3, {"heuristic": False, "kind": "grid", "metric": 1}),
])
def test_build_exact_small(eng, oracle, n, dim, kw):
    ida, kind = eng
    pc.check_build_exact(ida, oracle, n=n, dim=dim, seed=n, **kw)


@pytest.mark.parametrize("n,dim,kw", [
    (40, 3, {}), (70, 4, {"metric": 1}), (90, 5, {"keep_pruned": False}), (64, 300, {"ef_construction": 12}),
    (130, 3, {"ef_construction": 8}),        # full rows, tiny ef_construction: a neighbour's working set (65 x 65) dwarfs the new point's
    (80, 2, {"kind": "grid", "metric": 1, "ef_construction": 20}),
])
def test_build_exact_extend_candidates(eng, oracle, n, dim, kw):
    """Heuristic { extend_candidates: true } (core/lib.rs:648-664).  Upstream it deadlocks on the second insert (:649 read-locks
    a node write-locked at :438); its meaning without the locks is the oracle's restatement, and the GPU must match that byte
    for byte.  The schedule is sequential whatever max_batch says."""
    ida, kind = eng
    scale = 1 if kind != "gpu" else 6
    both = (("on-chip", {}), ("on-chip, smallest set the build allows (clamped to 256 ids) then bitmap", {"IDIST_TAB_LOG2": "7"}))
    pc.check_build_exact(ida, oracle, n=n * scale, dim=dim, seed=n, extend=True, max_batch=0, variants=both, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("n,dim,heur", [(1024, 2, True), (3000, 128, True), (2000, 300, True), (1200, 768, True),
                                        (1024, 2, False), (3000, 128, False), (1500, 300, False)])
def test_build_exact_gpu(engine_loader, oracle, n, dim, heur):
    ida = engine_loader("gpu")
    pc.check_build_exact(ida, oracle, n=n, dim=dim, seed=n, heuristic=heur)


@pytest.mark.gpu
@pytest.mark.parametrize("n,dim,efc", [(3000, 4, 60), (2500, 6, 100)])
def test_build_exact_gpu_squared_l2_lattice(engine_loader, oracle, n, dim, efc):
    """Integer lattices under squared L2: every selection is full of exact ties d(c_i, c_j) == d(c_i, q).  The Gram-matrix
    filter on the matrix cores (build_select_mfma_kernel) may decide none of them: they must all reach the canonical
    distance with the strict `<` of core/lib.rs:678, or the graph differs from the oracle's."""
    ida = engine_loader("gpu")
    pc.check_build_exact(ida, oracle, n=n, dim=dim, seed=n, kind="grid", ef_construction=efc)


@pytest.mark.gpu
def test_c1_isize_points_plumbing_gpu(engine_loader, oracle):
    """BASELINE config C1: 1k x 3-d isize points (README/colors.rs Point), k = 1.  isize coordinates in
    [0,255] are exact in f32, so the GPU engine answers it bit for bit like the reference's CPU path."""
    ida = engine_loader("gpu")
    rng = np.random.default_rng(1)
    pts = rng.integers(0, 256, size=(1000, 3)).astype(np.float32)
    q = rng.integers(0, 256, size=(200, 3)).astype(np.float32)
    h = pc.check_build_exact(ida, oracle, n=1000, dim=3, metric=1, seed=1, kind="uniform")  # same path, float data
    b = ida.Builder().metric(ida.METRIC_L2).max_batch(1)
    hi = ida.Hnsw.from_ordered_points(pts, b)
    oix = oracle.Index.build(pts, oracle.default_config(metric=1))
    assert np.array_equal(hi.into_parts()[0], oix.zero)
    got, want = hi.search_batch(q, ida.Search()), oix.search(q)
    assert np.array_equal(got.pid[:, 0], want.pid[:, 0])                       # k = 1
    assert np.array_equal(pc.bits(got.distance), pc.bits(want.dist))
    truth, td = hi.bruteforce(q, 1)
    assert np.mean(pc.bits(got.distance[:, 0]) == pc.bits(td[:, 0])) > 0.97    # nearest found (distance ties allowed)


@pytest.mark.gpu
def test_c2_full_size_properties_gpu(engine_loader, oracle):
    """BASELINE config C2 at full size (100k x 128 f32, k = 10, ef_search = 100): the oracle cannot
    build this in seconds, so check size-independent properties + oracle search parity on the
    GPU-built graph (the graph is an input of Hnsw::search)."""
    ida = engine_loader("gpu")
    rng = np.random.default_rng(2)
    pts = rng.random((100_000, 128), dtype=np.float32)
    q = rng.random((2000, 128), dtype=np.float32)
    q[:100] = pts[:100]
    h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
    got = h.search_batch(q, ida.Search(), counters=True)
    assert np.all(got.count == 100)
    assert np.all(got.distance[:, :-1] <= got.distance[:, 1:])                 # sortedness
    assert np.array_equal(got.pid[:100, 0], np.arange(100))                    # self query, distance 0
    assert np.all(got.distance[:100, 0] == 0)
    again = h.search_batch(q, ida.Search())
    assert np.array_equal(again.pid, got.pid)                                  # idempotence
    zero, layers = h.into_parts()
    assert [l.shape[0] for l in layers] == oracle.layer_sizes(100_000)[1:]
    oix = oracle.Index.from_arrays(pts, zero, layers, oracle.default_config())
    want = oix.search(q[:500], threads=8)
    assert np.array_equal(got.pid[:500], want.pid) and np.array_equal(got.counters[:500], want.counters)
    assert np.array_equal(pc.bits(got.distance[:500]), pc.bits(want.dist))
    truth, _ = h.bruteforce(q[:300], 10)
    assert pc.recall_at(got.pid[:300], truth, 10) > 0.5       # uniform 128-d data: high intrinsic dimension


@pytest.mark.parametrize("pipeline", ["1", "0"])
def test_build_batched(eng, oracle, monkeypatch, pipeline):
    # both schedules of a concurrent build: pipelined (descents of step k+1 overlap the updates of step k) and not
    ida, kind = eng
    pc.use_test_build(monkeypatch)                     # (schedule knobs exist in the test build only)
    monkeypatch.setenv("IDIST_BUILD_PIPELINE", pipeline)
    monkeypatch.setenv("IDIST_BUILD_CHECK", "1")      # pipelined: the two copies of the zero layer must agree at the end
    rec = pc.check_build_batched(ida, oracle, n=S(kind, 220, 30000), dim=S(kind, 4, 32), max_batch=S(kind, 4, 0),
                                 nq=S(kind, 20, 500))
    assert rec >= 0.95


def test_build_batched_is_schedule_independent(eng, oracle, monkeypatch):
    """A concurrent step's outcome must not depend on how its work is scheduled on the device: the same points
    built with different work-queue chunkings (which switches the update kernel's next-item prefetch on and
    off), with every update forced through the from-scratch kernel, and with the latency / throughput descent
    give byte-identical graphs."""
    ida, kind = eng
    pc.use_test_build(monkeypatch)                     # (schedule knobs exist in the test build only)
    rng = np.random.default_rng(4)
    pts = pc.gen_points(rng, S(kind, 150, 60000), S(kind, 6, 48), "lowrank" if kind == "gpu" else "uniform")
    b = ida.Builder().max_batch(S(kind, 16, 0))
    ref = None
    # (IDIST_BUILD_FILTER=0: descents without the reject filter — every distance in the log instead of bound-form entries)
    envs = [{}, {"IDIST_BUILD_CHUNK": "5"}, {"IDIST_BUILD_NO_FAST": "1"}, {"IDIST_LATENCY_NQ": "0"}, {"IDIST_BUILD_QUAD": "0"},
            {"IDIST_BUILD_FILTER": "0"}, {"IDIST_BUILD_QUAD": "0", "IDIST_BUILD_FILTER": "0"}, {"IDIST_BUILD_FILTER": "1"},
            {"IDIST_BUILD_QUAD": "0", "IDIST_BUILD_FILTER": "1"},
            {"IDIST_LATENCY_NQ": "0", "IDIST_WALK": "classic"}]
    if kind == "gpu":
        # (IDIST_BUILD_STREAMS=off: ONE descent stream with one queue head / visited region — the sequential steps of the growth phase
        #  must then stay on it, run_build's `seq1`)
        envs += [{"IDIST_BUILD_CHUNK": "1"}, {"IDIST_BUILD_CHUNK": "16"}, {"IDIST_LATENCY_NQ": "4000000000"}, {"IDIST_BUILD_STREAMS": "off"}]
    else:
        # which kernel selects for the new points (Gram matrix on MFMA / LDS tile), whether step B finds its distances in
        # the published log, how early the descents' visited set spills: none of it may show in the graph
        envs += [{"IDIST_BUILD_A2": "tile"}, {"IDIST_BUILD_NO_DLOG": "1"}, {"IDIST_TAB_LOG2": "7"}]
    for env in envs:
        with monkeypatch.context() as m:
            m.setenv("IDIST_BUILD_CHECK", "1")
            for k_, v in env.items():
                m.setenv(k_, v)
            zero, layers = ida.Hnsw.from_ordered_points(pts, b).into_parts()
        if ref is None:
            ref = (zero, layers)
            continue
        assert np.array_equal(zero, ref[0]), env
        assert all(np.array_equal(x, y) for x, y in zip(layers, ref[1])), env


@pytest.mark.gpu
def test_build_batched_matches_oracle_recall_gpu(engine_loader, oracle):
    """throughput-mode tier: recall@10 within noise of the oracle's parallel build."""
    ida = engine_loader("gpu")
    rng = np.random.default_rng(0)
    n, dim = 50000, 64
    pts = pc.gen_points(rng, n, dim, "lowrank")
    q = pc.gen_points(rng, 1000, dim, "lowrank")
    h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
    truth, _ = h.bruteforce(q, 10)
    got = h.search_batch(q, ida.Search())
    rec_gpu = pc.recall_at(got.pid, truth, 10)
    oix = oracle.Index.build(pts, oracle.default_config(), threads=8)
    rec_cpu = pc.recall_at(oix.search(q, threads=8).pid, truth, 10)
    assert rec_gpu >= rec_cpu - 0.02, (rec_gpu, rec_cpu)
    dz = (h.into_parts()[0] != pc.INVALID).sum(1).mean()
    do = (oix.zero != pc.INVALID).sum(1).mean()
    assert abs(dz - do) < 4, (dz, do)


@pytest.mark.gpu
def test_c3_full_size_properties_gpu(engine_loader, oracle):
    """BASELINE config C3 at full size (1M x 300 f32): build on the GPU, then (i) size-independent
    properties, (ii) the oracle searching the SAME exported graph must agree bit for bit, (iii) exact
    recall against the MFMA-filtered brute force."""
    ida = engine_loader("gpu")
    rng = np.random.default_rng(3)
    n, dim = 1_000_000, 300
    z = rng.standard_normal((n, 32), dtype=np.float32)
    a = np.random.default_rng(4242).standard_normal((32, dim), dtype=np.float32)
    pts = z @ a
    pts += 0.05 * rng.standard_normal((n, dim), dtype=np.float32)
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    q = pts[rng.integers(0, n, 64)] + 0.02 * rng.standard_normal((64, dim), dtype=np.float32)
    q = np.concatenate([pts[:64], q.astype(np.float32)])
    h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
    st = h.build_stats()
    assert st.n_updates_fast + st.n_updates_full == st.n_updates
    got = h.search_batch(q, ida.Search(), counters=True)
    assert np.all(got.count == 100) and np.all(got.distance[:, :-1] <= got.distance[:, 1:])
    assert np.array_equal(got.pid[:64, 0], np.arange(64)) and np.all(got.distance[:64, 0] == 0)   # self query
    zero, layers = h.into_parts()
    assert [l.shape[0] for l in layers] == oracle.layer_sizes(n)[1:]
    valid = zero != pc.INVALID
    assert np.all(valid[:, :-1] >= valid[:, 1:]) and np.all(zero[valid] < n) and valid.sum(1).min() >= 1
    oix = oracle.Index.from_arrays(pts, zero, layers, oracle.default_config())
    want = oix.search(q, threads=8)
    assert np.array_equal(got.pid, want.pid) and np.array_equal(got.counters, want.counters)
    assert np.array_equal(pc.bits(got.distance), pc.bits(want.dist))
    truth, _ = h.bruteforce(np.repeat(q, 2, axis=0), 10)        # 256 queries -> MFMA path
    assert pc.recall_at(got.pid, truth[::2], 10) > 0.9


@pytest.mark.gpu
def test_device_views_alias_index_gpu(engine_loader, oracle):
    """The zero-copy views used for RCCL replication alias the index's device buffers."""
    import subprocess
    import sys

    # own process: torch must initialise its HIP runtime before libidist does (see _capi.Lib)
    if "IDIST_DEVVIEW_CHILD" not in __import__("os").environ:
        env = dict(__import__("os").environ, IDIST_DEVVIEW_CHILD="1")
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", __file__ + "::test_device_views_alias_index_gpu"],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        return
    import torch

    torch.cuda.init()
    ida = engine_loader("gpu")
    from instant_distance_amd import dist as idd

    rng = np.random.default_rng(0)
    pts = rng.random((3000, 300), dtype=np.float32)
    h = ida.Hnsw.from_ordered_points(pts, ida.Builder())
    v_pts, v_zero, v_upper = idd.device_views(h, torch.device("cuda", 0))
    zero, layers = h.into_parts()
    assert np.array_equal(v_zero.cpu().numpy().view(np.uint32).reshape(-1, 64), zero)
    assert np.array_equal(v_upper.cpu().numpy().view(np.uint32).reshape(-1, 32), np.concatenate(layers))
    stride = h.info().row_stride
    rows = v_pts.cpu().numpy().view(np.float32).reshape(-1, stride)
    assert rows.shape[0] == 3000 and np.isclose(np.sort(rows[7])[-300:].sum(), np.sort(pts[7]).sum(), rtol=1e-5)
    # a replica filled through the views answers identically
    info = h.info()
    import ctypes as C
    from instant_distance_amd import _capi
    hh = C.c_void_p()
    cfg = ida.Builder()._config()
    ll = np.array(list(info.layer_len)[: info.n_upper], dtype=np.uint32)
    _capi.lib().check(_capi.lib().idist_index_alloc(3000, 300, C.byref(cfg), _capi.u32p(ll), info.n_upper, 0, C.byref(hh)))
    rep = ida.Hnsw(hh, pts, 100)
    for dst, src in zip(idd.device_views(rep, torch.device("cuda", 0)), (v_pts, v_zero, v_upper)):
        dst.copy_(src)
    torch.cuda.synchronize()
    q = rng.random((50, 300), dtype=np.float32)
    a, b = h.search_batch(q, ida.Search()), rep.search_batch(q, ida.Search())
    assert np.array_equal(a.pid, b.pid) and np.array_equal(pc.bits(a.distance), pc.bits(b.distance))


def test_bruteforce(eng, oracle):
    ida, kind = eng
    rng = np.random.default_rng(1)
    n, dim = S(kind, 200, 20000), S(kind, 10, 300)
    pts = pc.gen_points(rng, n, dim)
    q = pc.gen_points(rng, S(kind, 4, 64), dim)
    h = ida.Hnsw.from_parts(pts, np.full((n, 64), pc.INVALID, np.uint32), [], ida.Builder())
    pid, dist = h.bruteforce(q, 10)
    opid, odist = oracle.bruteforce(pts, q, 10, threads=4)
    assert np.array_equal(pid, opid) and np.array_equal(pc.bits(dist), pc.bits(odist))


def test_bruteforce_mfma_path_equals_scan(eng, oracle, monkeypatch):
    """BASELINE config C4's distance path: the f32-MFMA -2QP^T filter + canonical re-rank must return
    exactly what the scan kernel (and the oracle) return."""
    ida, kind = eng
    rng = np.random.default_rng(11)
    n, dim, nq, k = S(kind, 300, 60000), S(kind, 20, 300), S(kind, 130, 700), S(kind, 10, 10)
    pts = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    q[:3] = pts[[5, 17, n - 1]]
    pc.use_test_build(monkeypatch)                     # (IDIST_BRUTEFORCE / IDIST_BF_SAMPLE exist in the test build only)
    h = ida.Hnsw.from_parts(pts, np.full((n, 64), pc.INVALID, np.uint32), [], ida.Builder())
    monkeypatch.setenv("IDIST_BRUTEFORCE", "scan")
    p1, d1 = h.bruteforce(q, k)
    monkeypatch.setenv("IDIST_BRUTEFORCE", "mfma")
    monkeypatch.setenv("IDIST_BF_SAMPLE", str(S(kind, 64, 8192)))
    p2, d2 = h.bruteforce(q, k)
    assert np.array_equal(p1, p2) and np.array_equal(pc.bits(d1), pc.bits(d2))
    sub = slice(0, S(kind, nq, 40))
    op, od = oracle.bruteforce(pts, q[sub], k, threads=4)
    assert np.array_equal(p1[sub], op) and np.array_equal(pc.bits(d1[sub]), pc.bits(od))
    # tiny sample + large k: candidate lists overflow -> the call must still be exact (falls back to the scan)
    monkeypatch.setenv("IDIST_BF_SAMPLE", "1")
    p3, _ = h.bruteforce(q[:S(kind, 130, 300)], S(kind, 40, 100))
    monkeypatch.setenv("IDIST_BRUTEFORCE", "scan")
    p4, _ = h.bruteforce(q[:S(kind, 130, 300)], S(kind, 40, 100))
    assert np.array_equal(p3, p4)


def test_edge_cases(eng, oracle):
    ida, kind = eng
    s = ida.Search()
    # empty index (core/lib.rs:224-234, 359-361)
    h, ids = ida.Builder().build_hnsw(np.zeros((0, 4), np.float32))
    assert ids == [] and len(list(h.search(np.zeros(4, np.float32), s))) == 0
    # single point: only the entry point, never inserted (core/lib.rs:279-280)
    h, ids = ida.Builder().build_hnsw(np.ones((1, 4), np.float32))
    items = list(h.search(np.zeros(4, np.float32), s))
    assert ids == [0] and len(items) == 1 and items[0].pid == 0 and items[0].distance == 4.0
    # fewer points than ef_search: every point comes back once, sorted
    pts = np.random.default_rng(0).random((20, 3), dtype=np.float32)
    h, ids = ida.Builder().seed(1).build_hnsw(pts)
    items = list(h.search(pts[0], s))
    assert len(items) == 20 and items[0].distance == 0.0 and items[0].pid == ids[0]
    # no queries / ef_search = 0
    assert h.search_batch(np.zeros((0, 3), np.float32), s).pid.shape == (0, 100)
    h.set_ef_search(0)
    assert h.search_batch(pts[:2], s).count.tolist() == [0, 0]
    # dimension mismatch is an error, not UB
    with pytest.raises(TypeError):
        h.search_batch(np.zeros((1, 5), np.float32), s)
    # values shorter than points: the reference panics (core/lib.rs:148)
    with pytest.raises(IndexError):
        ida.Builder().build(pts, ["a"])


def test_import_validation_and_unsupported(eng, oracle):
    ida, kind = eng
    pts = np.random.default_rng(0).random((50, 4), dtype=np.float32)
    zero = np.full((50, 64), pc.INVALID, np.uint32)
    zero[3, 0] = 7
    zero[3, 1] = 7                          # duplicate: impossible in the reference (Visited)
    with pytest.raises(ida.IdistError) as e:
        ida.Hnsw.from_parts(pts, zero, [], ida.Builder())
    assert e.value.status == 5
    zero[3, 1] = 50                         # id >= n
    with pytest.raises(ida.IdistError) as e:
        ida.Hnsw.from_parts(pts, zero, [], ida.Builder())
    assert e.value.status == 5
    zero[3, 1] = pc.INVALID
    zero[3, 2] = 50                         # garbage after the first INVALID is never read (core/types.rs:183-187)
    ida.Hnsw.from_parts(pts, zero, [], ida.Builder())
    # Heuristic::extend_candidates deadlocks upstream; here it builds (test_build_exact_extend_candidates checks the graph)
    h, _ = ida.Builder().select_heuristic(ida.Heuristic(True, True)).build_hnsw(pts)
    assert len(h) == 50
    with pytest.raises(ida.IdistError) as e:
        ida.Builder().ef_search(5000).build_hnsw(pts)
    assert e.value.status == 1
    # select_heuristic(None) builds (the "simple" path of core/lib.rs:497-515)
    h, _ = ida.Builder().select_heuristic(None).seed(3).build_hnsw(pts)
    assert len(list(h.search(pts[0], ida.Search()))) == 50


def test_permutation_matches_oracle_restatement(eng, oracle):
    ida, kind = eng
    from instant_distance_amd import _capi
    import ctypes as C
    for seed, n in [(0, 1), (1, 10), (123456789, 1024), (2**63 + 5, 777)]:
        a = np.zeros(n, np.uint32); b = np.zeros(n, np.uint32)
        _capi.lib().check(_capi.lib().idist_permutation(C.c_uint64(seed), n, _capi.u32p(a), _capi.u32p(b)))
        oa, ob = oracle.permutation(seed, n)
        assert np.array_equal(a, oa) and np.array_equal(b, ob)
        assert sorted(a.tolist()) == list(range(n))


@pytest.mark.parametrize("order", ["forward", "reverse"])
def test_emulator_lane_schedule_independence(engine_loader, oracle, monkeypatch, order):
    """The lockstep emulator runs the lanes of a wave one after the other between collectives; results must not
    depend on that order (a cross-lane LDS hand-off without a barrier would), and no collective may be reached
    with lanes missing (STRICT).  Search (both variants), exact build and the heuristic=None build."""
    ida = engine_loader("emu")
    monkeypatch.setenv("IDIST_EMU_ORDER", order)
    monkeypatch.setenv("IDIST_EMU_STRICT", "1")
    pc.check_search_parity(ida, oracle, n=160, dim=12, ef_search=40, nq=3, seed=5)
    pc.check_search_parity(ida, oracle, n=120, dim=300, ef_search=100, nq=2, seed=6)
    pc.check_build_exact(ida, oracle, n=70, dim=6, seed=7)
    pc.check_build_exact(ida, oracle, n=80, dim=5, seed=8, heuristic=False)
    pc.check_build_batched(ida, oracle, n=130, dim=8, max_batch=0, nq=8, seed=9, min_recall=0.9)


@pytest.mark.gpu
def test_index_shared_by_threads_gpu(engine_loader, oracle):
    """`Hnsw::search(&self, …, &mut Search)` (core/lib.rs:352-356): the index is immutable and shareable, all
    mutable state lives in the caller's `Search`.  Several host threads, each with its own Search (= its own
    stream + scratch), query one index concurrently; every thread gets the oracle's answer.  (GPU only: the
    CPU emulator of tests/simt is single-threaded by design.)"""
    import threading

    ida, kind = engine_loader("gpu"), "gpu"
    pts, oix, _ = pc.oracle_graph(oracle, S(kind, 240, 20000), S(kind, 8, 96), "uniform", 0, 31, 1, 100)
    h = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, ida.Builder())
    rng = np.random.default_rng(32)
    n_thr = S(kind, 3, 6)
    qs = [pc.gen_points(rng, S(kind, 4, 300 + 700 * (i % 2)), pts.shape[1]) for i in range(n_thr)]   # narrow and wide batches
    wants = [oix.search(q, threads=1) for q in qs]
    errs = []

    def work(i):
        try:
            s = ida.Search()
            for _ in range(S(kind, 2, 5)):
                got = h.search_batch(qs[i], s, counters=True)
                pc.check_search_result(got, wants[i])
        except BaseException as e:  # noqa: BLE001
            errs.append((i, repr(e)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(n_thr)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


def test_builder_progress(eng, oracle):
    """Builder::progress (core/lib.rs:70-75): position never goes back, length = points.len(), ends at the length
    (`bar.finish()`), layers are reported top-down; the index is the same with or without a bar."""
    ida, kind = eng
    n = S(kind, 200, 200000)
    pts = pc.gen_points(np.random.default_rng(12), n, S(kind, 5, 64))
    seen = []
    b = ida.Builder().max_batch(S(kind, 1, 0)).progress(lambda done, total, layer: seen.append((done, total, layer)))
    h = ida.Hnsw.from_ordered_points(pts, b)
    assert seen and seen[-1][:2] == (n, n) and seen[-1][2] is None
    assert all(t == n for _, t, _ in seen)
    assert all(a[0] <= c[0] for a, c in zip(seen, seen[1:]))
    layers = [l for _, _, l in seen if l is not None]
    assert all(a >= c for a, c in zip(layers, layers[1:]))
    zero, _ = h.into_parts()
    zero2, _ = ida.Hnsw.from_ordered_points(pts, ida.Builder().max_batch(S(kind, 1, 0))).into_parts()
    assert np.array_equal(zero, zero2)
    # a watch that is never consumed must not leak into a later build of the same thread
    h2 = ida.Hnsw.from_ordered_points(pts[:50], ida.Builder())
    assert len(h2) == 50


def _fuzz_cases(kind, count, seed0):
    """Seeded random configurations: dimension classes (template / generic geometry, tails), data shapes with and
    without exact ties, both metrics, ef values around the 64-lane boundaries."""
    rng = np.random.default_rng(seed0)
    dims = [1, 2, 3, 5, 8, 12, 20, 31, 64, 100, 128, 129, 300, 304, 768] if kind == "gpu" else [1, 3, 5, 12, 20, 128, 300]
    out = []
    for i in range(count):
        dim = int(rng.choice(dims))
        kind_ = str(rng.choice(["uniform", "grid", "lowrank"] if dim >= 8 else ["uniform", "grid"]))
        n_hi = (6000 if dim <= 128 else 2500) if kind == "gpu" else 200
        n_lo = 400 if kind == "gpu" else 90
        out.append(dict(n=int(rng.integers(n_lo, n_hi)), dim=dim, kind=kind_, metric=int(rng.integers(0, 2)),
                        ef=int(rng.choice([1, 3, 17, 63, 64, 65, 100, 127, 128, 129, 200, 333])),
                        efc=int(rng.choice([8, 40, 64, 100, 130])), keep=bool(rng.integers(0, 2)), seed=seed0 * 1000 + i))
    return out


@pytest.mark.parametrize("case", range(3))
def test_fuzz_search_and_exact_build_emulated(engine_loader, oracle, case):
    ida = engine_loader("emu")
    c = _fuzz_cases("emu", 3, 77)[case]
    pc.check_search_parity(ida, oracle, n=c["n"], dim=c["dim"], ef_search=c["ef"], metric=c["metric"], kind=c["kind"],
                           nq=4, seed=c["seed"], ef_construction=c["efc"])
    pc.check_build_exact(ida, oracle, n=min(c["n"], 110), dim=c["dim"], metric=c["metric"], kind=c["kind"],
                         ef_construction=c["efc"], keep_pruned=c["keep"], seed=c["seed"] + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(14))
def test_fuzz_search_and_exact_build_gpu(engine_loader, oracle, case):
    ida = engine_loader("gpu")
    c = _fuzz_cases("gpu", 14, 78)[case]
    pc.check_search_parity(ida, oracle, n=c["n"], dim=c["dim"], ef_search=c["ef"], metric=c["metric"], kind=c["kind"],
                           nq=96, seed=c["seed"], ef_construction=c["efc"])
    pc.check_build_exact(ida, oracle, n=min(c["n"], 1500), dim=c["dim"], metric=c["metric"], kind=c["kind"],
                         ef_construction=c["efc"], keep_pruned=c["keep"], seed=c["seed"] + 1)


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(5))
def test_fuzz_concurrent_build_gpu(engine_loader, oracle, case):
    ida = engine_loader("gpu")
    c = _fuzz_cases("gpu", 5, 79)[case]
    n = c["n"] * (12 if c["dim"] <= 128 else 6)
    pc.check_build_concurrent_invariants(ida, oracle, n=n, dim=c["dim"], kind=c["kind"], metric=c["metric"],
                                         ef_construction=max(c["efc"], 40), keep_pruned=c["keep"], seed=c["seed"])


def test_concurrent_build_invariants_emulated(engine_loader, oracle):
    ida = engine_loader("emu")
    pc.check_build_concurrent_invariants(ida, oracle, n=330, dim=6, kind="grid", metric=1, max_batch=8, seed=5, nq=20, slack=0.08)


def test_tie_policy(eng, oracle):
    """> 64 un-expanded candidates exactly at the furthest distance (dense integer grids).  STRICT (default) enlarges
    the tie region by itself — the build is repeated, a host-pointer search batch is searched again — and stays
    bit-identical to the reference; DROP keeps the default region, goes on deterministically and flags the event."""
    ida, kind = eng
    rng = np.random.default_rng(3000002)
    n = S(kind, 420, 41640)
    pts = pc.gen_points(rng, n, S(kind, 3, 5), "grid")               # the case the concurrent-build fuzz found
    q = pts[: S(kind, 3, 20)] + np.float32(0.25)
    base = lambda: (ida.Builder().metric(1).ef_search(S(kind, 8, 100)).ef_construction(S(kind, 8, 64))   # noqa: E731
                    .max_batch(S(kind, 1, 0)))
    hs = ida.Hnsw.from_ordered_points(pts, base())                    # strict: succeeds, with a larger region if needed
    assert hs.build_stats().tie_overflow == 0
    escalated = hs.info().tie_capacity > 64
    if kind == "gpu":
        assert escalated                                              # 64 ties are not enough for this data
    h = ida.Hnsw.from_ordered_points(pts, base().tie_policy(ida.TIES_DROP))
    assert h.info().tie_capacity == 64 and h.build_stats().tie_overflow == (1 if escalated else 0)
    zero, layers = h.into_parts()
    if not escalated:                                                 # nothing dropped: the strict graph
        assert np.array_equal(zero, hs.into_parts()[0])
    zero2, _ = ida.Hnsw.from_ordered_points(pts, base().tie_policy(ida.TIES_DROP)).into_parts()
    assert np.array_equal(zero, zero2)
    ida.Hnsw.from_parts(pts, zero, layers, base().tie_policy(ida.TIES_DROP))        # row invariants hold
    s = ida.Search()
    got = h.search_batch(q, s)
    assert s.tie_overflowed() in (True, False)
    for i in range(len(q)):                                                         # the nearest points come back
        c = int(got.count[i])
        assert c >= 1 and np.all(got.distance[i, : c - 1] <= got.distance[i, 1:c])
        d0 = float(np.sqrt(np.min(np.sum((pts - q[i]) ** 2, axis=1))))
        assert abs(float(got.distance[i, 0]) - d0) < 1e-5
    # strict search on the oracle's graph: default region first, enlarged on demand, bit-identical to the oracle
    cfg = oracle.default_config(metric=1, ef_search=S(kind, 8, 100), ef_construction=S(kind, 8, 64))
    oix = oracle.Index.build(pts, cfg, threads=4)
    hb = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, ida.Builder().metric(1).ef_search(S(kind, 8, 100)))
    pc.check_search_result(hb.search_batch(q, ida.Search(), counters=True), oix.search(q))
    # ... and with the region requested up front
    hb2 = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, ida.Builder().metric(1).ef_search(S(kind, 8, 100)).tie_capacity(4096))
    pc.check_search_result(hb2.search_batch(q, ida.Search(), counters=True), oix.search(q))
    # data without mass ties is untouched by the policy: byte-identical graph, flag clear
    pts2 = pc.gen_points(rng, S(kind, 150, 4000), 6)
    a = ida.Hnsw.from_ordered_points(pts2, ida.Builder().max_batch(1))
    b = ida.Hnsw.from_ordered_points(pts2, ida.Builder().max_batch(1).tie_policy(ida.TIES_DROP))
    assert np.array_equal(a.into_parts()[0], b.into_parts()[0]) and b.build_stats().tie_overflow == 0


def test_strict_ties_enlarge_the_region_on_demand(eng, oracle):
    """The escalation machinery on data small enough for the oracle: with a deliberately tiny tie region (1 entry)
    integer-grid data overflows it at once; STRICT must still end with the reference's results (search batch searched
    again with 4x the region, build repeated with 8x), DROP must flag the overflow."""
    ida, kind = eng
    rng = np.random.default_rng(17)
    n, ef = S(kind, 260, 6000), S(kind, 12, 60)
    pts = rng.integers(0, 3, size=(n, 3)).astype(np.float32)
    q = rng.integers(0, 3, size=(S(kind, 12, 200), 3)).astype(np.float32) + np.float32(0.5)   # cell centres: many equal distances
    cfg = oracle.default_config(metric=1, ef_search=ef, ef_construction=ef)
    oix = oracle.Index.build(pts, cfg)
    want = oix.search(q)
    tiny = lambda: ida.Builder().metric(1).ef_search(ef).ef_construction(ef).max_batch(1).tie_capacity(1)   # noqa: E731
    # the data does overflow a 1-entry region
    hd = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, tiny().tie_policy(ida.TIES_DROP))
    sd = ida.Search()
    hd.search_batch(q, sd)
    assert sd.tie_overflowed()
    # strict search: same index config, the context enlarges its region until the batch goes through
    hs = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, tiny())
    for _, lat in pc.SEARCH_VARIANTS:
        with pc.search_variant(lat):
            pc.check_search_result(hs.search_batch(q, ida.Search(), counters=True), want)
    # strict build: repeated with a larger region, byte-identical to the oracle in the end
    hb = ida.Hnsw.from_ordered_points(pts, tiny())
    assert hb.info().tie_capacity > 1 and hb.build_stats().tie_overflow == 0
    zero, layers = hb.into_parts()
    assert np.array_equal(zero, oix.zero) and all(np.array_equal(x, y) for x, y in zip(layers, oix.layers))
    assert ida.Hnsw.from_ordered_points(pts, tiny().tie_policy(ida.TIES_DROP)).build_stats().tie_overflow == 1


def test_strict_ties_spill_to_hbm(eng, oracle, monkeypatch):
    """The reference's candidate heap is unbounded (core/lib.rs:564).  Ties that do not fit the LDS region — after it has
    grown to 4096 entries, or at once with IDIST_TIE_SPILL=1 as here — go to a per-slot bag in HBM and come back in
    (distance, pid) order: with a ONE-entry region on integer-grid data nearly every tie takes that road, and search (every
    walk variant) and exact build must still be the oracle's, bit for bit."""
    ida, kind = eng
    pc.use_test_build(monkeypatch)                     # (IDIST_TIE_SPILL exists in the test build only)
    monkeypatch.setenv("IDIST_TIE_SPILL", "1")
    rng = np.random.default_rng(23)
    n, ef = S(kind, 300, 8000), S(kind, 12, 60)
    pts = rng.integers(0, 3, size=(n, 3)).astype(np.float32)
    q = rng.integers(0, 3, size=(S(kind, 10, 200), 3)).astype(np.float32) + np.float32(0.5)   # cell centres: many equal distances
    cfg = oracle.default_config(metric=1, ef_search=ef, ef_construction=ef)
    oix = oracle.Index.build(pts, cfg)
    want = oix.search(q)
    tiny = lambda: ida.Builder().metric(1).ef_search(ef).ef_construction(ef).max_batch(1).tie_capacity(1)   # noqa: E731
    hs = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, tiny())
    for _, lat in pc.SEARCH_VARIANTS:
        with pc.search_variant(lat):
            s = ida.Search()
            pc.check_search_result(hs.search_batch(q, s, counters=True), want)
            pc.check_search_result(hs.search_batch(q[:3], s, counters=True), oix.search(q[:3]))   # the context keeps its bags
    hb = ida.Hnsw.from_ordered_points(pts, tiny())
    assert hb.info().tie_capacity == 1 and hb.build_stats().tie_overflow == 0     # no growth: the bags took the ties
    zero, layers = hb.into_parts()
    assert np.array_equal(zero, oix.zero) and all(np.array_equal(x, y) for x, y in zip(layers, oix.layers))
    # squared L2 on a 2-d grid, default region (64): the 5-d grid of test_tie_policy needs the growth + bags on the GPU only
    pts2 = pc.gen_points(rng, S(kind, 260, 6000), 2, "grid")
    o2 = oracle.Index.build(pts2, oracle.default_config(ef_search=S(kind, 20, 100)))
    h2 = ida.Hnsw.from_parts(pts2, o2.zero, o2.layers, ida.Builder().ef_search(S(kind, 20, 100)).tie_capacity(2))
    q2 = pts2[: S(kind, 6, 100)] + np.float32(0.5)
    pc.check_search_result(h2.search_batch(q2, ida.Search(), counters=True), o2.search(q2))


def test_concurrent_build_on_the_tie_bags(eng, oracle, monkeypatch):
    """A CONCURRENT (pipelined) build whose strict ties live in the HBM bags has one descent stream's worth of queue heads,
    visited bitmaps and bags: the sequential steps that open a layer must not run beside the next step's descents on them
    (run_build's `seq1`; round-5 advisor finding).  The graph must be the one the same schedule gives with a tie region large
    enough to need no bag — ties are handled bit-identically either way (core/lib.rs:564: the heap is unbounded) — and the two
    zero-layer copies of the pipeline must agree (IDIST_BUILD_CHECK)."""
    ida, kind = eng
    pc.use_test_build(monkeypatch)                     # (IDIST_TIE_SPILL / IDIST_BUILD_CHECK exist in the test build only)
    rng = np.random.default_rng(29)
    n, ef = S(kind, 330, 9000), S(kind, 12, 60)
    pts = pc.gen_points(rng, n, 3, "grid")             # 216 distinct integer points: duplicates and mass ties
    base = lambda: ida.Builder().metric(1).ef_search(ef).ef_construction(ef).max_batch(S(kind, 8, 0))   # noqa: E731
    monkeypatch.setenv("IDIST_BUILD_CHECK", "1")
    ref = ida.Hnsw.from_ordered_points(pts, base().tie_capacity(4096))
    zr, lr = ref.into_parts()
    monkeypatch.setenv("IDIST_TIE_SPILL", "1")
    for _ in range(S(kind, 1, 3)):                      # (a race would not show every time)
        hb = ida.Hnsw.from_ordered_points(pts, base().tie_capacity(1))
        assert hb.info().tie_capacity == 1 and hb.build_stats().tie_overflow == 0     # no growth: the bags took the ties
        zero, layers = hb.into_parts()
        assert np.array_equal(zero, zr) and all(np.array_equal(x, y) for x, y in zip(layers, lr))
    ida.Hnsw.from_parts(pts, zr, lr, base())           # idist_index_import validates every row


def _poison(rng, a, share):
    """NaN, +inf and -inf coordinates in `share` of the rows each (some rows get two kinds: inf - inf = NaN inside the distance)."""
    n, dim = a.shape
    for val in (np.nan, np.inf, -np.inf):
        rows = rng.choice(n, size=max(1, int(n * share)), replace=False)
        a[rows, rng.integers(0, dim, size=len(rows))] = val
    return a


def _canon_nan_bits(a):
    """OrderedFloat: all NaNs are equal (core/types.rs:229-234 via ordered-float) — x86 makes 0xFFC00000 out of inf - inf, the GPU
    stores its canonical 0x7FC00000; every other value must match bit for bit."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    return np.where(np.isnan(a), np.uint32(0x7FC00000), a.view(np.uint32))


@pytest.mark.parametrize("metric", [0, 1])
@pytest.mark.parametrize("dim", [5, 12, 300])
def test_non_finite_coordinates(eng, oracle, dim, metric):
    """NaN / +inf / -inf coordinates in points and queries.  The reference orders candidates by (OrderedFloat<f32>, PointId)
    (core/types.rs:229-234): NaN is the GREATEST distance and all NaNs tie (then the pid decides); +inf is an ordinary largest
    finite-side value.  Search on the oracle's graph (every walk variant) and the exact build must be the oracle's."""
    ida, kind = eng
    if kind == "emu" and dim == 300 and metric == 1:
        pytest.skip("300-d under the emulator once is enough")
    rng = np.random.default_rng(41 + dim + metric)
    n = S(kind, 140 if dim == 300 else 220, 2500)
    pts = _poison(rng, pc.gen_points(rng, n, dim), 0.04)
    q = _poison(rng, pc.gen_points(rng, S(kind, 10, 64), dim), 0.12)
    q[0] = pts[3]                                                      # a stored point (finite or not)
    ef = S(kind, 16, 100)
    cfg = oracle.default_config(metric=metric, ef_search=ef, ef_construction=S(kind, 20, 100))
    oix = oracle.Index.build(pts, cfg, threads=1)
    want = oix.search(q)
    assert np.isnan(want.dist[want.pid != pc.INVALID]).any()          # the case is live: NaN distances are among the answers
    b = ida.Builder().metric(metric).ef_search(ef).ef_construction(S(kind, 20, 100)).max_batch(1)
    h = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, b)
    search_vs = pc.pick_variants(pc.SEARCH_VARIANTS, dim + metric, 4)
    if kind == "gpu":                                                  # (every variant for every case costs the GPU suite minutes:
        search_vs = search_vs[:1] + search_vs[1 + (dim + metric) % 3::3]   #  the default + every third of the others, rotating by case)
    for _, lat in search_vs:
        with pc.search_variant(lat):
            got = h.search_batch(q, ida.Search(), counters=True)
        assert np.array_equal(got.count, want.count) and np.array_equal(got.pid, want.pid), lat
        assert np.array_equal(_canon_nan_bits(got.distance), _canon_nan_bits(want.dist)), lat
        assert np.array_equal(got.counters, want.counters), lat
    one = h.search_batch(q[:1], ida.Search(), counters=True)          # the scalar call (four waves per query)
    assert np.array_equal(one.pid, want.pid[:1]) and np.array_equal(one.counters, want.counters[:1])
    # the distance kernel on its own
    ids = rng.integers(0, n, size=(len(q), 40)).astype(np.uint32)
    wd = np.array([[oracle.distance(q[i], pts[j], metric) for j in ids[i]] for i in range(len(q))], dtype=np.float32)
    assert np.array_equal(_canon_nan_bits(h.distances(q, ids)), _canon_nan_bits(wd))
    # exact build: select_heuristic's `<` on OrderedFloat (core/lib.rs:676-679) with NaN / inf distances in the candidate sets
    build_vs = pc.pick_variants(pc.BUILD_VARIANTS, dim + metric, 2)
    if kind == "gpu":
        build_vs = build_vs[:1] + build_vs[1 + (dim + metric) % 3::3]
    for _, lat in build_vs:
        with pc.search_variant(lat):
            hb = ida.Hnsw.from_ordered_points(pts, b)
        zero, layers = hb.into_parts()
        assert np.array_equal(zero, oix.zero), lat
        assert all(np.array_equal(x, y) for x, y in zip(layers, oix.layers)), lat
        st = hb.build_stats()
        assert (st.n_dist, st.n_exp0, st.n_expU) == (oix.build_counters.n_dist, oix.build_counters.n_exp0, oix.build_counters.n_expU)
