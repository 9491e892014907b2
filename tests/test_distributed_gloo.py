"""N > 1 path on CPU: world_size 2, gloo.  Rank 0 builds, replicate_index() broadcasts the
index once, every rank searches its contiguous query shard, results are gathered and must be
identical to the oracle answering the whole batch.  Compute runs through the emulated kernels
(tests/simt) because this container has no GPU; on the GPU box the same code path uses nccl
(= RCCL) with zero-copy device views (instant-distance_amd/dist.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, emu_so, q):
    try:
        import torch.distributed as dist

        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import instant_distance_amd as ida
        from instant_distance_amd import _capi
        from instant_distance_amd import dist as idd

        _capi._singleton = _capi.Lib(emu_so)          # test-only engine swap (no GPU here)
        rng = np.random.default_rng(0)
        n, dim, nq = 260, 12, 31
        pts = rng.random((n, dim), dtype=np.float32)
        queries = rng.random((nq, dim), dtype=np.float32)
        builder = ida.Builder().max_batch(1).ef_search(40)
        hnsw = ida.Hnsw.from_ordered_points(pts, builder) if rank == 0 else None
        hnsw = idd.replicate_index(hnsw, ida.Builder().ef_search(40), src=0)
        lo, hi = idd.shard_range(nq, rank, world)
        r = hnsw.search_batch(queries[lo:hi], ida.Search(), counters=True)
        zero, layers = hnsw.into_parts()
        q.put((rank, lo, hi, r.pid, r.distance, r.count, zero))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback

        q.put((rank, "error", traceback.format_exc() + str(e)))


def test_shard_range_partitions():
    from instant_distance_amd.dist import shard_range

    for n in (0, 1, 7, 10000, 65536):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in parts) - min(hi - lo for lo, hi in parts) <= 1


@pytest.mark.timeout(600)
def test_replicate_and_shard_world2(oracle):
    import torch.multiprocessing as mp

    import engines

    emu_so = engines.build_emu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, emu_so, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(60)
    for g in got:
        assert g[1] != "error", g[2]
    got.sort(key=lambda g: g[0])
    # oracle on the whole batch
    rng = np.random.default_rng(0)
    pts = rng.random((260, 12), dtype=np.float32)
    queries = rng.random((31, 12), dtype=np.float32)
    oix = oracle.Index.build(pts, oracle.default_config(ef_search=40))
    want = oix.search(queries)
    assert np.array_equal(got[0][6], oix.zero) and np.array_equal(got[1][6], oix.zero)   # replica == source == oracle
    pid = np.concatenate([g[3] for g in got])
    dist_ = np.concatenate([g[4] for g in got])
    cnt = np.concatenate([g[5] for g in got])
    assert (got[0][1], got[0][2], got[1][1], got[1][2]) == (0, 15, 15, 31)
    assert np.array_equal(pid, want.pid) and np.array_equal(cnt, want.count)
    assert np.array_equal(dist_.view(np.uint32), want.dist.view(np.uint32))


@pytest.mark.gpu
def test_rccl_single_rank_replicate_gpu():
    """backend nccl (= RCCL) on the real GPU with one rank: meta broadcast + bulk broadcast straight out of the
    library-owned device buffers (zero-copy views).  Own process: torch must bring up its HIP runtime first."""
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "nccl_selftest.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "nccl selftest ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
