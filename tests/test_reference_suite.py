"""The reference's own tests and examples, replayed through the host API mirror
(instant_distance_amd.Builder / HnswMap / Search) — CPU run uses the emulated kernels
(tests/simt), `-m gpu` run uses the MI355X.

  instant-distance/tests/all.rs:11-39   map
  instant-distance/tests/all.rs:41-88   random_heuristic, random_simple
  instant-distance/examples/colors.rs   nearest colour
  instant-distance-py/test/test.py      self query on 1024 x 300
"""
import numpy as np
import pytest

from engines import engine_params


@pytest.fixture(params=engine_params())
def ida(request, engine_loader):
    return engine_loader(request.param)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 11])
def test_map(ida, seed):
    pts = np.array([[i, i] for i in range(5)], dtype=np.float32)
    values = ["zero", "one", "two", "three", "four"]
    m = ida.Builder.default().seed(seed).metric(ida.METRIC_L2).build(pts, values)
    search = ida.Search()
    items = list(m.search(np.array([2.0, 2.0], dtype=np.float32), search))
    assert len(items) == 5
    for i, item in enumerate(items):
        if i == 0:
            assert np.float32(item.distance) == np.float32(0.0) and item.value == "two"
        elif i in (1, 2):
            assert np.float32(item.distance) == np.float32(1.4142135) and item.value in ("one", "three")
        else:
            assert np.float32(item.distance) == np.float32(2.828427) and item.value in ("zero", "four")


def _randomized(ida, builder, seed, n):
    rng = np.random.default_rng(seed)
    pts = rng.random((n, 2), dtype=np.float32)
    query = rng.random(2, dtype=np.float32)
    hnsw, pids = builder.seed(seed).metric(ida.METRIC_L2).build_hnsw(pts)
    search = ida.Search()
    results = list(hnsw.search(query, search))
    assert len(results) >= min(100, n)
    d = np.sqrt(((pts - query) ** 2).sum(1, dtype=np.float32))
    forced = set(pids[i] for i in np.argsort(d, kind="stable")[:100])
    found = set(it.pid for it in results[:100])
    return len(forced & found)


def test_random_heuristic(ida, sizes):
    n = sizes["random_n"]
    recall = _randomized(ida, ida.Builder.default(), 123456789, n)
    assert recall > 97, recall            # tests/all.rs:45


def test_random_simple(ida, sizes):
    n = sizes["random_n"]
    recall = _randomized(ida, ida.Builder.default().select_heuristic(None), 987654321, n)
    assert recall > 90, recall            # tests/all.rs:52


def test_colors(ida):
    pts = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255]], dtype=np.float32)
    values = ["red", "green", "blue"]
    m = ida.Builder.default().seed(7).metric(ida.METRIC_L2).build(pts, values)
    s = ida.Search()
    assert next(m.search(np.array([204, 85, 0], np.float32), s)).value == "red"
    first = next(m.search(np.array([163, 193, 173], np.float32), s))
    assert first.value == "green"
    assert np.float32(first.distance) == np.sqrt(np.float32(163 ** 2 + 62 ** 2 + 173 ** 2))


def test_self_query_300d(ida, sizes):
    n = sizes["self_n"]
    rng = np.random.default_rng(5)
    emb = rng.random((n, 300), dtype=np.float32)
    words = [f"w{i}" for i in range(n)]
    m = ida.Builder.default().seed(99).build(emb, words)
    s = ida.Search()
    chosen = min(123, n - 1)
    first = next(m.search(emb[chosen], s))
    assert first.value == words[chosen] and first.distance == 0.0
