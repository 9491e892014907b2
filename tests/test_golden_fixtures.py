"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the pinned
oracle): the oracle must still reproduce them, and so must the engine (CPU: emulated kernels, GPU: -m gpu)."""
import glob
import os

import numpy as np
import pytest

from engines import engine_params

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def _load(path):
    z = np.load(path)
    n, dim, nq, metric, heur, ef = [int(x) for x in z["meta"]]
    layers = [z[f"layer{i}"] for i in range(int(z["n_layers"]))]
    return z, layers, metric, heur, ef


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_oracle_reproduces_golden(oracle, path):
    z, layers, metric, heur, ef = _load(path)
    ix = oracle.Index.build(z["points"], oracle.default_config(metric=metric, has_heuristic=heur, ef_search=ef))
    assert np.array_equal(ix.zero, z["zero"])
    assert all(np.array_equal(a, b) for a, b in zip(ix.layers, layers)) and len(ix.layers) == len(layers)
    r = ix.search(z["queries"])
    assert np.array_equal(r.pid, z["pid"]) and np.array_equal(r.dist.view(np.uint32), z["dist_bits"])
    assert np.array_equal(r.count, z["count"]) and np.array_equal(r.counters, z["counters"])


@pytest.fixture(params=engine_params())
def ida(request, engine_loader):
    return engine_loader(request.param)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_engine_reproduces_golden(ida, path):
    z, layers, metric, heur, ef = _load(path)
    b = ida.Builder().metric(metric).ef_search(ef).max_batch(1).select_heuristic(ida.Heuristic() if heur else None)
    h = ida.Hnsw.from_ordered_points(z["points"], b)
    zero, got_layers = h.into_parts()
    assert np.array_equal(zero, z["zero"])
    assert len(got_layers) == len(layers) and all(np.array_equal(a, b_) for a, b_ in zip(got_layers, layers))
    r = h.search_batch(z["queries"], ida.Search(), counters=True)
    assert np.array_equal(r.pid, z["pid"]) and np.array_equal(r.distance.view(np.uint32), z["dist_bits"])
    assert np.array_equal(r.count, z["count"]) and np.array_equal(r.counters, z["counters"])
