"""BASELINE.json configurations C4 (1M x 768-d, 65,536 batched queries) and C5 (10M x 768-d, one GPU's replica) at FULL
size on the MI355X, through the same code that prints their bench lines (`bench.py --config C4|C5 --check`):

  * the index is built on the GPU from device-resident points (Builder::build, core/lib.rs:209-345);
  * the CPU oracle searches the SAME exported graph (the graph is an input of Hnsw::search, core/lib.rs:352-383) for a
    512-query sample at ef_search 100 and 200 (and the timed one): ids, order, counts, distance bits and the work counters
    {n_dist, n_exp0, n_expU} must be identical — `parity` and `cpu_baseline` of the JSON line;
  * size-independent properties of the full batch and of the graph — `checks`: every query returns ef results, sorted,
    ids unique, a second Search gives the same bytes, stored points find themselves at distance 0, layer sizes are the
    reference's f32 table, rows are prefix-valid with ids < n and no self link;
  * recall@10 against the exact ground truth (-2QP^T filter on MFMA + canonical re-rank) reaches the target.

Own process per configuration: torch brings the HIP runtime up before libidist (see _capi.Lib), and 31 GB of points
go away with the process."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_config(name, extra=()):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", name, "--check", "--steps", "3", "--warmup", "1",
           "--threads", "", "--cpu-sample", "512", "--cpu-build-sample", "0", "--no-traffic", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def assert_config(out, n, dim, nq):
    cfg = out["config"]
    assert (cfg["n"], cfg["dim"], cfg["queries_per_gpu"]) == (n, dim, nq)
    # oracle parity on the same graph: ids / order / counts / distance bits / work counters
    par = out["parity"]
    assert par["queries"] >= 512 and par["ef100"] and par["ef200"] and par["all_identical"], par
    cpu = out["cpu_baseline"]
    assert cpu["ids_identical_to_gpu"] and cpu["ids_distance_bits_counts_and_work_counters_identical_to_gpu"], cpu
    # size-independent properties
    ck = out["checks"]
    for key in ("count_is_ef", "sorted_nearest_first", "ids_unique_per_query", "idempotent", "self_query_first_at_distance_0",
                "layer_sizes_match_reference", "rows_prefix_valid", "row_ids_in_range", "no_self_links"):
        assert ck[key] is True, (key, ck)
    assert ck["min_degree"] >= 1
    # the operating point: smallest ef of the sweep with recall@10 >= 0.95 against the exact ground truth
    assert cfg["recall_target_met"] and cfg["recall_at_10"] >= 0.95, cfg
    assert set(cfg["ef_sweep_recall"]) >= {"100", "200"}
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and 0.0 < rf["frac"] < 1.0 and rf["kernel_ms_avg"] > 0
    # the reject filter is at work at these shapes — a silent fall-back to the unfiltered walk would halve the rate
    assert (rf["reject_filter"]["rejected_share"] or 0) > 0.8, rf["reject_filter"]
    assert (out["build"]["reject_filter"]["rejected_share"] or 0) > 0.8, out["build"]["reject_filter"]
    assert out["build"]["n_updates_memoised"] + out["build"]["n_updates_full"] == out["build"]["n_updates"]


@pytest.mark.gpu
def test_c4_full_size_properties_gpu():
    out = run_config("C4")
    assert_config(out, 1_000_000, 768, 65_536)


@pytest.mark.gpu
def test_c5_full_size_properties_gpu():
    out = run_config("C5")
    assert_config(out, 10_000_000, 768, 65_536)
    assert len(out["checks"]) and out["config"]["ef_search"] in (100, 200, 400)


@pytest.mark.gpu
def test_bench_under_torchrun_one_rank_gpu():
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the N > 1 control flow on the real backend with a world of one
    (nccl init on the device, replicate_index over RCCL out of the library's device buffers, ef agreement, barriers around the timed
    steps, replica digest + rank-0 oracle check, the --rccl-child process after the process group is gone).  Small index: what is
    tested is that every one of those lines runs on hardware; tests/test_distributed_gloo.py runs them at world sizes 2 and 3."""
    import socket

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "C2", "--points", "50000", "--nq", "2000",
           "--steps", "3", "--warmup", "1", "--threads", "", "--cpu-sample", "512", "--cpu-build-sample", "0", "--no-traffic", "--check"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["replication"].startswith("rank 0 built, nccl broadcast")
    assert out["config"]["replicate_bytes"] >= 50000 * (128 * 4 + 256)
    rc = out["replica_check"]
    assert rc["all_ranks_identical_to_rank0"] and rc["rank0_identical_to_oracle"] and rc["queries"] == 512
    child = out["config"]["replicate_rccl_in_process"]
    assert child.get("last_replica_answers_identical_to_root") is True and child["peer_ranks_exited_before_launch"], child
    assert out["parity"]["all_identical"] and out["cpu_baseline"]["ids_distance_bits_counts_and_work_counters_identical_to_gpu"]
    assert all(v is True for k, v in out["checks"].items() if k != "min_degree")
