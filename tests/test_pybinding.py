"""The wheel-compatible surface (instant_distance_amd.pybinding = the reference's `instant_distance`
module, instant-distance-py/src/lib.rs) replayed on the reference's own Python test
(instant-distance-py/test/test.py) plus load/dump round trips and a byte-level check of the bincode
layout (restated from the serde derives: parity unpinned, no sample file exists)."""
import struct

import numpy as np
import pytest

from engines import engine_params


@pytest.fixture(params=engine_params())
def idp(request, engine_loader):
    engine_loader(request.param)
    import instant_distance_amd.pybinding as instant_distance

    return instant_distance, (48 if request.param == "emu" else 1024)


def test_hnsw(idp):
    # instant-distance-py/test/test.py:4-12
    instant_distance, n = idp
    rng = np.random.default_rng(0)
    points = [[float(x) for x in rng.random(300)] for _ in range(n)]
    config = instant_distance.Config()
    (hnsw, ids) = instant_distance.Hnsw.build(points, config)
    assert sorted(ids) == list(range(n))
    p = [float(x) for x in rng.random(300)]
    search = instant_distance.Search()
    hnsw.search(p, search)
    got = list(search)
    assert len(got) == min(n, config.ef_search)
    assert all(got[i].distance <= got[i + 1].distance for i in range(len(got) - 1))
    assert got[0].value is None and "Item(" in repr(got[0])


def test_hnsw_map(idp):
    # instant-distance-py/test/test.py:15-35
    instant_distance, n = idp
    the_chosen_one = min(123, n - 1)
    rng = np.random.default_rng(1)
    embeddings = [[float(x) for x in rng.random(300)] for _ in range(n)]
    values = [f"word{i}" for i in range(n)]
    config = instant_distance.Config()
    hnsw_map = instant_distance.HnswMap.build(embeddings, values, config)
    search = instant_distance.Search()
    hnsw_map.search(embeddings[the_chosen_one], search)
    first = next(search)
    assert first.value == values[the_chosen_one] and first.distance == 0.0


def test_short_points_are_zero_padded_and_long_rejected(idp):
    instant_distance, _ = idp
    pts = [[1.0, 2.0], [3.0], [0.5, 0.5, 0.5]]          # py/lib.rs:363-376
    hnsw, ids = instant_distance.Hnsw.build(pts, instant_distance.Config())
    s = instant_distance.Search()
    hnsw.search([3.0], s)
    first = next(s)
    assert first.distance == 0.0 and first.pid == ids[1]
    with pytest.raises(TypeError):
        instant_distance.Hnsw.build([[0.0] * 301], instant_distance.Config())


def test_dump_load_roundtrip_and_layout(idp, tmp_path):
    instant_distance, n = idp
    n = min(n, 200)
    rng = np.random.default_rng(2)
    emb = rng.random((n, 300), dtype=np.float32)
    vals = [f"v{i}" for i in range(n)]
    cfg = instant_distance.Config()
    cfg.seed = 7
    cfg.ef_search = 33
    m = instant_distance.HnswMap.build(emb, vals, cfg)
    f = str(tmp_path / "map.idx")
    m.dump(f)
    m2 = instant_distance.HnswMap.load(f)
    s1, s2 = instant_distance.Search(), instant_distance.Search()
    for q in emb[:5]:
        m.search(q, s1)
        m2.search(q, s2)
        a, b = list(s1), list(s2)
        assert [(x.pid, x.distance, x.value) for x in a] == [(x.pid, x.distance, x.value) for x in b]
        assert len(a) == min(33, n)
    # bincode 1.3 layout: u64 ef_search | u64 n | n*300 f32 | u64 n | n*64 u32 | u64 layers ... | u64 n | (u32 0, u64 len, bytes)*
    raw = open(f, "rb").read()
    assert struct.unpack_from("<QQ", raw, 0) == (33, n)
    off = 16 + n * 1200
    assert struct.unpack_from("<Q", raw, off)[0] == n
    zero, layers = m._inner.hnsw.into_parts()
    assert np.array_equal(np.frombuffer(raw, "<u4", n * 64, off + 8).reshape(n, 64), zero)
    off += 8 + n * 256
    assert struct.unpack_from("<Q", raw, off)[0] == len(layers)
    off += 8
    for l in layers:
        assert struct.unpack_from("<Q", raw, off)[0] == l.shape[0]
        off += 8 + l.shape[0] * 128
    assert struct.unpack_from("<Q", raw, off)[0] == n
    assert struct.unpack_from("<IQ", raw, off + 8) == (0, len(m._inner.values[0].encode()))
    # plain Hnsw file
    h, ids = instant_distance.Hnsw.build(emb, cfg)
    g = str(tmp_path / "hnsw.idx")
    h.dump(g)
    h2 = instant_distance.Hnsw.load(g)
    h.search(emb[3], s1)
    h2.search(emb[3], s2)
    assert [(x.pid, x.distance) for x in s1] == [(x.pid, x.distance) for x in s2]
    with pytest.raises(ValueError):
        open(g, "r+b").truncate(100)
        instant_distance.Hnsw.load(g)
