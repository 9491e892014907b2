"""Several GPUs behind the C ABI: idist_replicate + idist_search_batch_sharded, and the
lifetime rules of a search context (Search scratch grows on demand; a context is bound to ONE index).

The test box has one GPU (the emulator: one device), so the replicas land on the devices that exist plus a
second copy on device 0 — the copies, the per-shard host threads and the block partition are the same code
whatever the device list is.  Bar: shard-concat == single-GPU result == oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import parity_cases as pc
from engines import engine_params


@pytest.fixture(params=engine_params())
def eng(request, engine_loader):
    return engine_loader(request.param), request.param


def S(kind, emu, gpu):
    return gpu if kind == "gpu" else emu


def test_replicate_and_sharded_search_match_single_gpu_and_oracle(eng, oracle):
    ida, kind = eng
    from instant_distance_amd import _capi

    rng = np.random.default_rng(5)
    n, dim, nq = S(kind, 300, 20000), S(kind, 12, 128), S(kind, 23, 1001)
    pts = rng.random((n, dim), dtype=np.float32)
    q = rng.random((nq, dim), dtype=np.float32)
    cfg = oracle.default_config(ef_search=40)
    oix = oracle.Index.build(pts, cfg)
    want = oix.search(q)
    h = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, ida.Builder().ef_search(40))
    single = h.search_batch(q, ida.Search(), counters=True)
    pc.check_search_result(single, want)
    devices = list(range(_capi.lib().device_count())) + [0]          # every GPU there is, plus a second copy on GPU 0
    for rccl in (True, False):                                        # one RCCL broadcast per buffer / peer copies from the root
        reps = h.replicate(devices, rccl=rccl)
        assert len(reps) == len(devices)
        for r, d in zip(reps, devices):
            assert r.info().device == d and r.info().n == n
            z, layers = r.into_parts()
            assert np.array_equal(z, oix.zero) and all(np.array_equal(a, b) for a, b in zip(layers, oix.layers))
            # the point rows travelled too: the replica answers like the root, bit for bit
            pc.check_search_result(r.search_batch(q[:16], ida.Search(), counters=True), oix.search(q[:16]))
        if rccl:
            assert h.last_replicate_seconds >= 0.0
    for shards in (reps, reps[:1], [h] + reps):                      # any mix of replicas (the root is one too)
        got = ida.Hnsw.search_batch_sharded(shards, [ida.Search() for _ in shards], q, counters=True)
        pc.check_search_result(got, want)
    # fewer queries than shards: empty ranges are skipped
    got = ida.Hnsw.search_batch_sharded([h] + reps, [ida.Search() for _ in range(len(reps) + 1)], q[:2], counters=True)
    assert np.array_equal(got.pid, want.pid[:2]) and np.array_equal(got.counters, want.counters[:2])
    # Item.point works on a replica: the host copy of the points is shared
    s = ida.Search()
    it = next(iter(reps[-1].search(pts[7], s)))
    assert it.pid == 7 and it.distance == 0.0 and np.array_equal(it.point, pts[7])


def test_sharded_search_rejects_bad_shard_sets(eng, oracle):
    ida, kind = eng
    rng = np.random.default_rng(6)
    pts = rng.random((200, 8), dtype=np.float32)
    h = ida.Hnsw.from_ordered_points(pts, ida.Builder().max_batch(1))
    other = ida.Hnsw.from_ordered_points(pts[:150], ida.Builder().max_batch(1))
    s = ida.Search()
    with pytest.raises(ida.IdistError) as e:                          # one context for two shards: they would race
        ida.Hnsw.search_batch_sharded([h, h], [s, s], pts[:4])
    assert e.value.status == 1
    with pytest.raises(ida.IdistError) as e:                          # not replicas of one index
        ida.Hnsw.search_batch_sharded([h, other], [ida.Search(), ida.Search()], pts[:4])
    assert e.value.status == 1


def test_context_is_bound_to_its_index_not_to_an_address(eng, oracle):
    """a context created for index A must be rejected for ANY other index — also one that lives at the
    address A had (free A, build B: `new` very likely hands the address out again)."""
    ida, kind = eng
    from instant_distance_amd import _capi

    L = _capi.lib()
    rng = np.random.default_rng(7)
    a = ida.Hnsw.from_ordered_points(rng.random((120, 6), dtype=np.float32), ida.Builder().max_batch(1))
    ctx = C.c_void_p()
    L.check(L.idist_search_ctx_new(a._h, 0, C.byref(ctx)))
    addr_a = a._h.value
    del a                                                             # frees index A; ctx outlives it
    seen_same_address = False
    for _ in range(S(kind, 4, 8)):
        b = ida.Hnsw.from_ordered_points(rng.random((S(kind, 150, 4000), 6), dtype=np.float32), ida.Builder())
        seen_same_address |= b._h.value == addr_a
        q = np.zeros((1, 6), dtype=np.float32)
        pid = np.zeros((1, 100), dtype=np.uint32)
        dd = np.zeros((1, 100), dtype=np.float32)
        cnt = np.zeros(1, dtype=np.uint32)
        st = L.idist_search_batch(b._h, ctx, _capi.f32p(q), 1, _capi.u32p(pid), _capi.f32p(dd), _capi.u32p(cnt), None)
        assert st == 1 and b"does not belong" in L.idist_last_error()
        if seen_same_address:
            break
        del b
    L.idist_search_ctx_free(ctx)                                      # freeing a context after its index is fine


def test_search_scratch_grows_on_demand(eng, oracle):
    """Search::default() is cheap (core/lib.rs:767-778, scratch sized on first use :363): a fresh context backs one
    query slot; batches make it grow; results do not depend on the slot count."""
    ida, kind = eng
    rng = np.random.default_rng(8)
    n, dim = S(kind, 260, 8000), S(kind, 8, 64)
    pts = rng.random((n, dim), dtype=np.float32)
    q = rng.random((S(kind, 21, 700), dim), dtype=np.float32)
    oix = oracle.Index.build(pts, oracle.default_config(ef_search=30))
    want = oix.search(q)
    h = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, ida.Builder().ef_search(30))
    s = ida.Search()                                                  # slots = 0: grows
    for hi in (1, 3, len(q), 2):
        got = h.search_batch(q[:hi], s, counters=True)
        assert np.array_equal(got.pid, want.pid[:hi]) and np.array_equal(got.counters, want.counters[:hi])
    for slots in (1, 2, 5):                                           # fixed slot counts: fewer slots than queries
        got = h.search_batch(q, ida.Search(slots), counters=True)
        pc.check_search_result(got, want)
    s2 = ida.Search()                                                 # Vec::reserve on the scratch: backed up front, same answers
    s2.reserve(h, len(q))
    pc.check_search_result(h.search_batch(q, s2, counters=True), want)


def test_narrow_host_batches_zero_copy_equals_staged(eng, oracle, monkeypatch):
    """Host-pointer batches of up to ~100 queries cross PCIe through the context's pinned, device-mapped buffer (the kernel
    reads the queries and writes the results itself, the work-queue head runs on from launch to launch); wider ones and
    IDIST_NO_ZERO_COPY=1 take the staged copies.  Same results either way, call after call on one context, and == oracle."""
    ida, kind = eng
    pc.use_test_build(monkeypatch)                     # (IDIST_NO_ZERO_COPY exists in the test build only)
    rng = np.random.default_rng(5)
    n, dim = S(kind, 220, 6000), S(kind, 6, 48)
    pts = rng.random((n, dim), dtype=np.float32)
    oix = oracle.Index.build(pts, oracle.default_config(ef_search=30))
    h = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, ida.Builder().ef_search(30))
    q = rng.random((S(kind, 9, 300), dim), dtype=np.float32)
    want = oix.search(q)
    for env in ({}, {"IDIST_NO_ZERO_COPY": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = ida.Search()
        for width in (1, 2, 3, S(kind, 4, 150)):                       # the last width does not fit the zero-copy buffer on the GPU
            for lo in range(0, len(q) - width + 1, width):
                got = h.search_batch(q[lo:lo + width], s, counters=True)
                assert np.array_equal(got.pid, want.pid[lo:lo + width]) and np.array_equal(got.count, want.count[lo:lo + width])
                assert np.array_equal(got.distance.view(np.uint32), want.dist[lo:lo + width].view(np.uint32))
                assert np.array_equal(got.counters, want.counters[lo:lo + width])
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.gpu
@pytest.mark.parametrize("combine", ["1", "0"])
def test_scalar_calls_from_many_threads_gpu(engine_loader, oracle, monkeypatch, combine):
    """The reference's concurrency model (core/lib.rs:352-356): many threads, one `Search` each, scalar `Hnsw::search`
    calls on ONE shared index.  Beyond eight launches in flight a call rides along in another thread's launch
    (idist_combine.hpp); every caller must still get exactly its own query's answer — ids, order, counts, distance bits,
    work counters — and so it must with every call launching by itself (IDIST_COMBINE=0)."""
    import threading

    monkeypatch.setenv("IDIST_COMBINE", combine)
    ida = engine_loader("gpu")
    pts, oix, _ = pc.oracle_graph(oracle, 20000, 96, "uniform", 0, 61, 1, 100)
    h = ida.Hnsw.from_parts(pts, oix.zero, oix.layers, ida.Builder())
    rng = np.random.default_rng(62)
    n_thr, calls = 24, 40
    q = pc.gen_points(rng, n_thr * calls, pts.shape[1])
    want = oix.search(q, threads=8)
    errs = []
    gate = threading.Barrier(n_thr)

    def work(t):
        try:
            s = ida.Search()
            h.search_batch(q[:1], s)                        # the context's buffers come into being
            gate.wait()
            for i in range(calls):
                j = t * calls + i
                got = h.search_batch(q[j:j + 1], s, counters=True)
                assert np.array_equal(got.pid[0], want.pid[j]) and got.count[0] == want.count[j], (t, i)
                assert np.array_equal(pc.bits(got.distance[0]), pc.bits(want.dist[j])), (t, i)
                assert np.array_equal(got.counters[0], want.counters[j]), (t, i)
        except BaseException as e:  # noqa: BLE001
            errs.append((t, repr(e)))
            try:
                gate.abort()
            except Exception:  # noqa: BLE001
                pass

    ts = [threading.Thread(target=work, args=(t,)) for t in range(n_thr)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a scalar call never returned"
    assert not errs, errs[:3]

