"""N > 1 path on CPU (gloo; compute through the emulated kernels of tests/simt because this container has no GPU; on the GPU box
the same code paths use nccl = RCCL with zero-copy device views, instant-distance_amd/dist.py):

  * world size 2: rank 0 builds, replicate_index() broadcasts the index once, every rank searches its contiguous query shard,
    results are gathered and must be identical to the oracle answering the whole batch;
  * world sizes 2 and 3: bench.py's own N > 1 control flow — `bench.run_bench` over gloo / cpu Jobs — the weak-scaling line,
    --config C5's split batch with uneven shards, and a deliberately corrupted replica that must stop every rank;
  * -m gpu: backend nccl with one rank on the real device (scripts/nccl_selftest.py)."""
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


# ---- bench.py's N > 1 control flow (the code the 8-GPU lease runs), world sizes 2 and 3 ----
def _bench_worker(rank, world, port, emu_so, q, scenario, argv):
    """One rank of `bench.run_bench` on a gloo / cpu Job around the emulator build of the product sources: the same functions,
    in the same order, as `torchrun ... bench.py --gpus N` runs them on nccl / cuda."""
    try:
        import torch
        import torch.distributed as dist

        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import instant_distance_amd as ida
        from instant_distance_amd import _capi
        from instant_distance_amd import dist as idd

        import bench

        _capi._singleton = _capi.Lib(emu_so)          # test-only engine swap (no GPU here)
        job = bench.Job(torch, rank, world, 0, torch.device("cpu"), dist)   # the emulator has one device: every rank uses device 0
        args = bench.parse_args(argv)
        if scenario == "bench":
            out = bench.run_bench(job, args)
            q.put((rank, "ok", out))
        elif scenario == "corrupt":
            # rank 0 builds, everyone gets a replica, then the LAST rank's replica loses most links of its zero layer: the digest must
            # fail on EVERY rank (SystemExit), not only on the rank that holds the bad copy
            n, dim = args.n, args.dim
            builder = ida.Builder().max_batch(args.max_batch)
            hnsw = bench.phase_build(job, ida, builder, n, dim)[0] if rank == 0 else None
            hnsw, _, _ = bench.phase_replicate(job, idd, hnsw, builder)
            ok_before = bench.phase_replica_check(job, ida, hnsw, args.nq, dim, 40)[0]
            if rank == world - 1:
                zero, layers = hnsw.into_parts()
                # every row of the zero layer keeps only its first three links: whatever rows the sample's walks expand, the
                # walks (and their work counters) differ
                zero[:, 3:] = 0xFFFFFFFF
                hnsw = ida.Hnsw.from_parts(hnsw.points, zero, layers, ida.Builder().ef_search(40))
            try:
                bench.phase_replica_check(job, ida, hnsw, args.nq, dim, 40)
                q.put((rank, "ok", {"before": ok_before, "raised": None}))
            except SystemExit as e:
                q.put((rank, "ok", {"before": ok_before, "raised": str(e)}))
        elif scenario == "nohost":
            # the source built from device-resident points and holds no host copy: a host transport cannot replicate it, and EVERY
            # rank must learn that before the first bulk broadcast (the defect bench's first gloo execution found: the source
            # skipped an array the others were waiting for)
            pts = bench.synth(torch, args.n, args.dim, 1, job.dev)
            hnsw = ida.Hnsw.from_device_points(pts.data_ptr(), args.n, args.dim, ida.Builder().max_batch(1)) if rank == 0 else None
            try:
                idd.replicate_index(hnsw, ida.Builder(), src=0)
                q.put((rank, "ok", {"raised": None}))
            except RuntimeError as e:
                q.put((rank, "ok", {"raised": str(e)}))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as e:  # noqa: BLE001
        import traceback

        q.put((rank, "error", traceback.format_exc() + str(e)))


def _run_world(world, scenario, argv, timeout=400):
    import torch.multiprocessing as mp

    import engines

    emu_so = engines.build_emu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, emu_so, q, scenario, argv)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time

    got, t0 = [], time.time()
    try:
        while len(got) < world:
            try:
                got.append(q.get(timeout=2))
            except queue.Empty:
                # a rank that died without reporting (an abort inside a collective) must fail the test now, not at the timeout
                dead = [p for p in procs if p.exitcode not in (None, 0)]
                assert not dead, f"rank process(es) died: exit codes {[p.exitcode for p in procs]}; reports so far: {got}"
                assert time.time() - t0 < timeout, f"no report after {timeout} s; reports so far: {got}"
    finally:
        for p in procs:
            p.join(30 if len(got) == world else 0.1)
            if p.is_alive():
                p.kill()
    for g in got:
        assert g[1] != "error", g[2]
    return [g[2] for g in sorted(got, key=lambda g: g[0])]


TOY = ["--n", "300", "--dim", "12", "--steps", "2", "--warmup", "1", "--threads", "", "--no-traffic", "--max-batch", "1",
       "--gt-queries", "0", "--parity-queries", "8"]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3])
def test_bench_weak_scaling_flow(world):
    """`bench.py --gpus N` (weak: every rank its own nq queries): the line rank 0 prints, produced at world sizes 2 and 3."""
    outs = _run_world(world, "bench", TOY + ["--gpus", str(world), "--config", "C3", "--nq", "9", "--check"])
    out = outs[0]
    assert all(o is None for o in outs[1:])                            # ONE line, from rank 0
    assert out["n_gpus"] == world and out["scaling"] == "weak" and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["queries_per_gpu"] == 9 and out["value"] > 0 and out["ms_per_step"] > 0
    assert abs(out["value"] - 9 * world / (out["ms_per_step"] * 1e-3)) <= 1e-3 * out["value"]      # whole-job rate
    assert out["config"]["replication"].startswith("rank 0 built, gloo broadcast") and out["config"]["replicate_bytes"] > 300 * 12 * 4
    rc = out["replica_check"]
    assert rc["all_ranks_identical_to_rank0"] and rc["rank0_identical_to_oracle"] and rc["queries"] == 9 * world
    assert out["parity"]["all_identical"] and out["cpu_baseline"] is None   # CPU timing is rank 0 at N = 1 only
    assert out["roofline"]["alg_bytes_per_launch"] > 0 and out["roofline"]["traffic"] is None
    assert all(out["checks"].values())


@pytest.mark.timeout(900)
def test_bench_split_batch_uneven_shards_world3():
    """--config C5's path at toy size: ONE global batch split over the ranks ("strong"), 31 queries over 3 ranks (10 / 10 / 11),
    ef chosen by the ladder and agreed by all_reduce."""
    outs = _run_world(3, "bench", TOY + ["--gpus", "3", "--config", "C5", "--nq", "31"])
    out = outs[0]
    assert out["scaling"] == "strong" and out["n_gpus"] == 3 and out["config"]["queries_per_gpu"] == 10
    assert abs(out["value"] - 31 / (out["ms_per_step"] * 1e-3)) <= 1e-3 * out["value"]
    assert out["config"]["ef_search"] in (100, 200, 400, 800) and set(out["config"]["ef_sweep_recall"]) >= {"100", "200"}
    assert out["replica_check"]["all_ranks_identical_to_rank0"] and out["replica_check"]["rank0_identical_to_oracle"]
    assert out["replica_check"]["queries"] == 31


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3])
def test_bench_corrupted_replica_fails_digest_on_every_rank(world):
    res = _run_world(world, "corrupt", TOY + ["--gpus", str(world), "--nq", "24"])
    for rank, r in enumerate(res):
        assert r["before"]["all_ranks_identical_to_rank0"]             # the honest replicas pass ...
        assert r["raised"] is not None and "differ from rank 0" in r["raised"], (rank, r)   # ... the bad one stops every rank


@pytest.mark.timeout(600)
def test_host_transport_without_host_points_stops_every_rank():
    res = _run_world(2, "nohost", TOY + ["--gpus", "2"])
    for r in res:
        assert r["raised"] is not None and "host copy of the points" in r["raised"], r


def test_bench_wait_for_peers():
    import subprocess

    import bench

    p = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(1.0)"])
    gone, waited = bench.wait_for_peers([os.getpid(), p.pid], timeout=30)
    p.wait()
    assert gone and 0.2 <= waited < 30
    p = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(20)"])
    gone, _ = bench.wait_for_peers([p.pid], timeout=0.5)
    p.kill(); p.wait()
    assert not gone


@pytest.mark.gpu
def test_rccl_single_rank_replicate_gpu():
    """backend nccl (= RCCL) on the real GPU with one rank: meta broadcast + bulk broadcast straight out of the
    library-owned device buffers (zero-copy views).  Own process: torch must bring up its HIP runtime first."""
    import subprocess

    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "nccl_selftest.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "nccl selftest ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
