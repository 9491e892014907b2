#!/usr/bin/env python
"""bench.py — queries/sec at recall@10 >= 0.95 on 1M x 300-d f32 (BASELINE.json config C3, the default),
plus index build points/sec, on N MI355X GPUs of one node.  `--config C2|C4|C5` runs the other BASELINE
configurations through the same code and prints the same JSON shape.

A "step" = one pass of Hnsw::search over this rank's batch of synthetic queries, inputs and
outputs resident in HBM.  N > 1 (launched by torch.distributed.run, one rank per GPU): rank 0
builds the index, RCCL broadcasts it once over xGMI, every rank searches its own query
shard, no collective inside the timed region.

Structure: the measurement is a set of functions over a `Job` (the torch backend that carries the collectives, the
device that holds the tensors, the libidist that answers) — phase_build, phase_replicate, phase_queries, phase_choose_ef,
phase_timed, phase_replica_check, rank0_report, composed by run_bench().  main() makes the nccl / cuda Job;
tests/test_distributed_gloo.py makes gloo / cpu Jobs around the emulator build of the product sources and runs the same
functions at world sizes 2 and 3, so the N > 1 control flow has executed before an 8-GPU lease runs it.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     HBM bound: the bytes one launch requests by the kernel's own counters — f32 rows of the pushed candidates the
               reject filter did not turn down + compact rows of the candidates it examined + adjacency rows +
               results; = SURVEY 8(d)'s B_q = n_dist*4*D + n_exp0*256 + n_expU*128 + 8*ef when nothing is filtered, and that
               figure is kept beside it as `survey_8d` — / average kernel duration measured with HIP events on the launch
               stream; `traffic` = fabric bytes per launch from two child passes of this command under rocprofv3 --pmc,
               calibrated in the same pass on both request patterns (N = 1).
  cpu_baseline the CPU oracle (restated reference, NOT the Rust crate) searching the SAME graph
               on the host cores for a bounded query sample (rank 0 at N = 1 only).
  parity       (N = 1) the oracle's answers for a query sample at ef_search 100 / 200 / the timed one compared with
               the GPU's: ids, order, counts, distance bits, work counters.
  replica_check (N > 1, and under torch.distributed.run at N = 1) every rank's answers for a common sample against rank 0's
               (digest), rank 0's against the oracle.
  checks       (--check) size-independent properties of the results and of the built graph; tests/ assert on them.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per host thread of the scalar-call measurement: a process-wide runtime setting, the HOST's to make (libidist
# does not touch the environment) and read once when the HIP runtime starts — so before `import torch`
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md chip table: 8.0 TB/s spec (6.29 TB/s measured copy)


def source_stamp():
    """The commit the tree under test was built from: `.build_commit` (written by __graft_entry__.build() / scripts/stamp.py,
    which travels to the GPU box — .git does not), else `git rev-parse` where a repository is at hand."""
    try:
        return open(os.path.join(ROOT, ".build_commit")).read().strip()
    except OSError:
        pass
    try:
        import subprocess
        sha = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
        dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--untracked-files=no"], capture_output=True, text=True, timeout=10).stdout.strip()
        return (sha + ("+dirty" if dirty else "")) if sha else None
    except Exception:  # noqa: BLE001
        return None

# BASELINE.json configs.  nq: queries per GPU per step ("weak") or of the whole job ("strong": C5's
# 65,536-query batch is split over the GPUs).  gtq: queries with exact ground truth.  cpu_build: prefix the CPU oracle builds.
CONFIGS = {
    "C2": dict(n=100_000, dim=128, nq=10_000, split=False, gtq=0, cpu_build=100_000,
               what="C2: 100k x 128-d f32, k=10, ef_search=100"),
    "C3": dict(n=1_000_000, dim=300, nq=10_000, split=False, gtq=0, cpu_build=100_000,
               what="C3: 1M x 300-d f32 (fastText-shape), Builder::build + 10k-query search"),
    "C4": dict(n=1_000_000, dim=768, nq=65_536, split=False, gtq=4096, cpu_build=40_000,
               what="C4: 1M x 768-d f32, 65,536 batched queries (ground truth by the -2QP^T MFMA filter + canonical re-rank)"),
    "C5": dict(n=10_000_000, dim=768, nq=65_536, split=True, gtq=2048, cpu_build=40_000,
               what="C5: 10M x 768-d f32, index replicated, 65,536-query batch sharded across the GPUs"),
}


def synth(torch, n, dim, seed, device, latent=32):
    """'fastText-shape' synthetic rows: z~N(0,I_32) A_{32xD} + 0.05 N(0,I_D), L2-normalised."""
    g = torch.Generator(device=device)
    g.manual_seed(4242)
    a = torch.randn(latent, dim, generator=g, device=device, dtype=torch.float32)
    g.manual_seed(seed)
    out = torch.empty(n, dim, device=device, dtype=torch.float32)
    step = 1 << 18
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        z = torch.randn(hi - lo, latent, generator=g, device=device, dtype=torch.float32)
        x = z @ a + 0.05 * torch.randn(hi - lo, dim, generator=g, device=device, dtype=torch.float32)
        out[lo:hi] = x / x.norm(dim=1, keepdim=True)
    return out


def effective_cores():
    """Host cores this process may actually use: min(affinity, cgroup CPU quota)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    cores = min(cores, max(1, quota // period))
            break
        except Exception:  # noqa: BLE001
            continue
    return cores


def measure_traffic(args, n, dim, nq, ef, mix=None):
    """roofline.traffic, measured in this session: two child passes of THIS command under `rocprofv3 --pmc` (FETCH_SIZE and
    WRITE_SIZE cannot share a pass: TCC has four counter slots; PMC passes serialise kernels, so they cannot run inside the
    timed process).  FETCH_SIZE is corrected with a factor calibrated IN THE SAME PASS on the same access pattern with a known
    byte count (MI355X_MICROARCH.md §HBM: gfx950 tallies 128-B requests at 64 B; calibrate in your own pattern).  Returns
    (dict | None, reason)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="idist_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", ctr, "-d", d, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "--traffic-child", "--config", args.config,
                   "--n", str(n), "--dim", str(dim), "--nq", str(nq), "--ef", str(ef), "--max-batch", str(args.max_batch)]
            r = subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                return None, f"{ctr} pass failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"
            known = [int(l.split()[1]) for l in r.stdout.splitlines() if l.startswith("known_read_bytes_per_launch")]
            known_c = [int(l.split()[1]) for l in r.stdout.splitlines() if l.startswith("known_compact_read_bytes_per_launch")]
            dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if not dbs:
                return None, f"{ctr} pass left no results database"
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (ctr,)).fetchall()
            sk = [v for k, v in rows if "search_kernel" in k]
            cal = [v for k, v in rows if "distance_batch_kernel" in k]
            cal_c = [v for k, v in rows if "filter_bound_kernel" in k]
            if not sk:
                return None, f"{ctr} pass saw no search_kernel dispatch"
            full = [v for v in sk if v >= 0.5 * max(sk)]                    # the full-batch launches
            got[ctr] = {"kb": sum(full) / len(full), "launches": len(full), "calib_kb": cal[-1] if cal else None, "known": known[-1] if known else None,
                        "calib_c_kb": cal_c[-1] if cal_c else None, "known_c": known_c[-1] if known_c else None}
        except Exception as e:  # noqa: BLE001
            return None, f"{ctr} pass: {e!r}"[:300]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    f = got["FETCH_SIZE"]
    if not f["calib_kb"] or not f["known"]:
        return None, "no calibration dispatch in the FETCH_SIZE pass"
    factor_rows = f["known"] / (f["calib_kb"] * 1024.0)
    factor_compact = f["known_c"] / (f["calib_c_kb"] * 1024.0) if f.get("calib_c_kb") and f.get("known_c") else None
    # The filtered walk mixes two request patterns, and FETCH_SIZE under-reports them differently (a request is tallied at 64 B
    # whatever its size): correct the reported bytes with the factor of the kernel's OWN byte mix — its f32 rows and adjacency rows
    # at the f32-gather factor, its compact rows at theirs (mix = the kernel's counters, rank0_report).
    factor = factor_rows
    if mix and factor_compact and mix.get("compact", 0) > 0:
        total = mix["rows"] + mix["compact"]
        factor = total / (mix["rows"] / factor_rows + mix["compact"] / factor_compact)
    fetch = f["kb"] * 1024.0 * factor
    write = got["WRITE_SIZE"]["kb"] * 1024.0
    return {"bytes_per_launch": int(fetch + write), "fetch_bytes_reported": int(f["kb"] * 1024), "fetch_correction_factor": round(factor, 4),
            "fetch_correction_factor_f32_rows": round(factor_rows, 4),
            "fetch_correction_factor_compact_rows": round(factor_compact, 4) if factor_compact else None,
            "fetch_bytes_corrected": int(fetch), "write_bytes_reported": int(write), "launches_averaged": f["launches"],
            "calibration": "distance_batch_kernel / filter_bound_kernel over a random permutation of all rows of the same index in the same "
                           f"PMC pass: f32 rows {f['known']} B known, {int(f['calib_kb'] * 1024)} B reported"
                           + (f"; compact rows {f['known_c']} B known, {int(f['calib_c_kb'] * 1024)} B reported; the launch's own byte mix weights the two"
                              if factor_compact else ""),
            "how": "two child passes of this command under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, this box, this session; "
                   "WRITE_SIZE uncalibrated (9 MB of 70 GB); fabric bytes (Infinity-Cache hits included)"}, None


def rccl_child(args, cfgd, torch):
    """One process, N devices: build on device 0, idist_replicate_rccl to devices 1 .. N-1 (and once to device 0 itself, so that
    the call is exercised on a single-GPU box too), then the last replica must answer a query sample exactly like the root."""
    import instant_distance_amd as ida

    ndev = min(args.rccl_child, torch.cuda.device_count())
    n, dim = args.n or cfgd["n"], args.dim or cfgd["dim"]
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    d_pts = synth(torch, n, dim, 123456789, dev)
    torch.cuda.synchronize()
    root = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder().max_batch(args.max_batch))
    root.set_ef_search(args.ef or 100)
    dests = list(range(1, ndev)) or [0]
    t0 = time.perf_counter()
    reps = root.replicate(dests, rccl=True)
    wall = time.perf_counter() - t0
    secs = root.last_replicate_seconds
    q = synth(torch, 512, dim, 123456790, dev).cpu().numpy()
    a = root.search_batch(q, ida.Search(), counters=True)
    reps[-1].set_ef_search(args.ef or 100)
    b = reps[-1].search_batch(q, ida.Search(), counters=True)
    same = bool(np.array_equal(a.pid, b.pid) and np.array_equal(a.distance.view(np.uint32), b.distance.view(np.uint32)) and
                np.array_equal(a.counters, b.counters))
    info = root.info()
    nbytes = int(n * info.row_stride * 4 + n * 256 + sum(list(info.layer_len)[: info.n_upper]) * 128)
    print(json.dumps({"seconds": round(secs, 3), "wall_seconds_with_communicator_setup": round(wall, 3), "destinations": dests, "bytes": nbytes,
                      "GBps_per_destination": round(nbytes / max(secs, 1e-9) / 1e9, 2), "last_replica_answers_identical_to_root": same,
                      "note": "idist_replicate_rccl from ONE process (ncclCommInitAll + one grouped ncclBroadcast per buffer), measured after the "
                              "per-rank processes exited"}), flush=True)


def scalar_calls(hnsw, ida, q_host, n_threads, calls):
    """The reference's own concurrency model (core/lib.rs:352-356): T host threads share ONE index, each owns a
    `Search` and issues scalar `Hnsw::search` calls (idist_search_batch with nq = 1, host pointers).  Returns the
    aggregate calls/s.  The C entry point is called directly with preallocated buffers (ctypes drops the GIL for the
    duration of the call), so Python's share per call is a few microseconds."""
    import ctypes as C

    from instant_distance_amd import _capi

    L = _capi.lib()
    ef, nq = hnsw._ef_search, q_host.shape[0]
    searches = [ida.Search() for _ in range(n_threads)]
    ctxs = [s._bind(hnsw) for s in searches]
    gate = threading.Barrier(n_threads + 1)
    errs = []

    def work(t):
        pid = np.empty(ef, np.uint32); dd = np.empty(ef, np.float32); cnt = np.zeros(1, np.uint32)
        pp, pd, pc = _capi.u32p(pid), _capi.f32p(dd), _capi.u32p(cnt)
        qs = [_capi.f32p(q_host[(t * calls + i) % nq]) for i in range(calls)]
        fn, h, ctx = L.idist_search_batch, hnsw._h, ctxs[t]
        rc = fn(h, ctx, qs[0], 1, pp, pd, pc, None)            # first call: the context's buffers come into being
        gate.wait()
        for i in range(min(30, calls)):                        # untimed: the contexts behind combined launches come into being too
            rc |= fn(h, ctx, qs[i], 1, pp, pd, pc, None)
        gate.wait()
        for i in range(calls):
            rc |= fn(h, ctx, qs[i], 1, pp, pd, pc, None)
        gate.wait()
        if rc:
            errs.append(rc)

    ts = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for t in ts:
        t.start()
    gate.wait()
    gate.wait()
    t0 = time.perf_counter()
    gate.wait()
    dt = time.perf_counter() - t0
    for t in ts:
        t.join()
    if errs:
        raise RuntimeError(f"idist_search_batch failed in a thread: status {errs}")
    kts = [s.kernel_times_ms(32) for s in searches]                # HIP events of the launches that served the last calls
    kts = [k.mean() for k in kts if len(k)]
    scalar_calls.kernel_ms = float(np.mean(kts)) if kts else None
    return n_threads * calls / dt


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS), help="BASELINE.json configuration (default C3, the metric's)")
    # (--points: torch.distributed.run's own parser rejects a bare `--n` after the script name as an ambiguous abbreviation)
    ap.add_argument("--n", "--points", dest="n", type=int, default=0, help="override the configuration's point count")
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--nq", type=int, default=0, help="queries per GPU per step (C5: of the whole job)")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--ef", type=int, default=0, help="ef_search; 0 = smallest of 100/200/400/800 reaching the recall target")
    ap.add_argument("--recall-target", type=float, default=0.95)
    ap.add_argument("--gt-queries", type=int, default=-1, help="queries with exact ground truth (0 = all; MFMA -2QP^T filter + exact re-rank)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="queries for the CPU baseline (0 = auto, ~15 s)")
    ap.add_argument("--parity-queries", type=int, default=512, help="queries the oracle answers at every ef of the parity object")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-build-sample", type=int, default=-1, help="points of the prefix the CPU oracle builds (threaded); 0 = skip")
    ap.add_argument("--check", action="store_true", help="add the `checks` object (size-independent properties; used by tests/)")
    ap.add_argument("--threads", default="1,4,16,64", help="host-thread counts of the scalar-call measurement ('' = skip)")
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes that fill roofline.traffic (N = 1 only)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)   # internal: the profiled child of measure_traffic()
    ap.add_argument("--no-inproc-rccl", action="store_true", help="N > 1: skip the in-process idist_replicate_rccl measurement after the run")
    ap.add_argument("--rccl-child", type=int, default=0, help=argparse.SUPPRESS)        # internal: devices of the in-process replication child
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="--gpus N > 1 started without a launcher: seconds the N ranks get")
    ap.add_argument("--collective-timeout", type=float, default=900.0, help="bound of every torch.distributed wait (C5 broadcasts 33.8 GB)")
    return ap.parse_args(argv)


class Job:
    """One rank's surroundings: the torch backend that carries the collectives, the device that holds the tensors, and (through
    instant_distance_amd._capi) the libidist that answers.  main() makes the nccl (= RCCL) / cuda:LOCAL_RANK one.
    tests/test_distributed_gloo.py makes gloo / cpu ones around the emulator build of the same sources, so every line of the
    N > 1 control flow below (replication, ef agreement, barriers around the timed steps, max over ranks, replica digest, rank-0
    oracle check) has run at world sizes 2 and 3 before an 8-GPU lease runs it.  `dist` is None for a plain `python bench.py`."""

    def __init__(self, torch, rank=0, world=1, local_rank=0, dev=None, dist=None):
        self.torch, self.rank, self.world, self.local_rank, self.dist = torch, rank, world, local_rank, dist
        self.dev = dev if dev is not None else torch.device("cuda", local_rank)
        self.cuda = self.dev.type == "cuda"
        # tensors handed to collectives live where the backend wants them: in HBM for nccl, on the host for gloo
        self.cdev = self.dev if (dist is not None and dist.get_backend() == "nccl") else torch.device("cpu")

    @property
    def multi(self):
        return self.dist is not None

    def sync(self):
        if self.cuda:
            self.torch.cuda.synchronize()

    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream if self.cuda else 0

    def barrier(self):
        if self.multi:
            self.dist.barrier()

    def reduce(self, value, dtype, op):
        """all_reduce of one scalar; the identity without a process group"""
        if not self.multi:
            return value
        t = self.torch.tensor([value], dtype=dtype, device=self.cdev)
        self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
        return t.item()

    def bcast_i64(self, value, src=0):
        t = self.torch.tensor([value], dtype=self.torch.int64, device=self.cdev)
        self.dist.broadcast(t, src=src)
        return int(t.item())

    def gather_i64(self, value):
        ts = [self.torch.zeros(1, dtype=self.torch.int64, device=self.cdev) for _ in range(self.world)]
        self.dist.all_gather(ts, self.torch.tensor([value], dtype=self.torch.int64, device=self.cdev))
        return [int(t.item()) for t in ts]


def phase_build(job, ida, builder, n, dim):
    """Rank 0: synthetic points in HBM, Builder::build on the GPU.  -> (hnsw, d_pts, build object)"""
    torch = job.torch
    d_pts = synth(torch, n, dim, 123456789, job.dev)
    job.sync()
    t0 = time.time()
    # (a host-transport job — gloo, tests — replicates from the host copy of the points; RCCL broadcasts the device buffers)
    hnsw = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, builder, host_points=None if job.cuda else d_pts.numpy())
    t_build = time.time() - t0
    st = hnsw.build_stats()
    secs = max(st.seconds, 1e-9)
    build = {"points_per_s": round(n / secs, 1), "device_seconds": round(st.seconds, 3),
             "wall_seconds": round(t_build, 3), "ef_construction": 100, "batches": int(st.n_batches),
             "n_dist": int(st.n_dist), "n_sel_pairs": int(st.n_sel_pairs), "n_updates": int(st.n_updates),
             "n_updates_memoised": int(st.n_updates_fast), "n_updates_full": int(st.n_updates_full),
             "n_heur_rows": int(st.n_heur_rows),
             "n_sel_pairs_note": "candidate pairs of select_heuristic decided here (by the Gram-matrix filter, a memoised "
                                 "verdict or the canonical distance) — NOT the reference's early-exit count of distance calls"}
    # Bytes the build's algorithm moves AS EXECUTED HERE: the descents (B_q with ef_construction on the partial
    # graph), every point row fetched for select_heuristic / the neighbour re-selections (n_heur_rows; pairwise
    # reuse is on chip, memoised verdicts fetch nothing), and the adjacency rows read + rewritten.
    # Round 6: the descents of rows >= 256 floats look a candidate up in the compact copy of its row first and fetch
    # the f32 row only of those the filter does not turn down: `examined` compact rows + (n_dist - rejected) f32 rows.
    rest = int(st.n_exp0 * 256 + st.n_expU * 128 + st.n_heur_rows * 4 * dim + st.n_updates * 512 + n * 256)
    survey_ab = int(st.n_dist * 4 * dim) + rest
    ab = int((st.n_dist - st.n_filter_rejected) * 4 * dim + st.n_filter_examined * st.filter_row_bytes) + rest
    build["reject_filter"] = {"examined": int(st.n_filter_examined), "rejected": int(st.n_filter_rejected),
                              "rejected_share": round(st.n_filter_rejected / st.n_filter_examined, 4) if st.n_filter_examined else None,
                              "compact_row_bytes": int(st.filter_row_bytes)}
    build["roofline"] = {"bound": "hbm", "achieved": round(ab / secs / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(ab / secs / 1e9 / HBM_PEAK_GBPS, 4), "alg_bytes": ab,
                         "alg_bytes_one_f32_row_per_push": survey_ab, "frac_one_f32_row_per_push": round(survey_ab / secs / 1e9 / HBM_PEAK_GBPS, 4),
                         "note": "all build kernels together over the build's device time; bytes the kernels request by their own counters "
                                 "(descents: f32 rows of the candidates the reject filter did not turn down + compact rows of those it examined); "
                                 "per-kernel PMC bytes: profiles/"}
    return hnsw, d_pts, build


def phase_replicate(job, idd, hnsw, builder):
    """One broadcast of rank 0's index into every other rank's device buffers, bracketed by barriers.
    -> (this rank's replica, seconds, config entries).  A failure fails the run (no silent per-rank rebuild: that curve would
    not exercise the replication path)."""
    job.sync()
    job.barrier()
    t0 = time.time()
    hnsw = idd.replicate_index(hnsw, builder, src=0)
    job.sync()
    job.barrier()
    t_rep = time.time() - t0
    rep_bytes = idd.device_buffer_bytes(hnsw)
    rep = {"replicate_bytes": rep_bytes, "replicate_GBps_per_destination": round(rep_bytes / max(t_rep, 1e-9) / 1e9, 2),
           "replicate_note": "one broadcast tree over xGMI (7 links x ~153 GB/s per GPU); seconds include the first RCCL call's set-up"}
    return hnsw, t_rep, rep


def phase_queries(job, idd, nq_total, dim):
    """Held-out draws, seed+1, generated identically on every rank; rank r keeps its contiguous block of the global batch."""
    d_q_all = synth(job.torch, nq_total, dim, 123456790, job.dev)
    lo, hi = idd.shard_range(nq_total, job.rank, job.world)
    d_q = d_q_all[lo:hi].contiguous()
    return d_q, lo, hi


class Runner:
    """The step: one pass of Hnsw::search over this rank's resident batch, results into resident buffers."""

    def __init__(self, job, ida, hnsw, d_q):
        self.job, self.ida, self.hnsw, self.d_q, self.nq = job, ida, hnsw, d_q, int(d_q.shape[0])
        self.search = ida.Search()

    def alloc_out(self, ef, nq=None):
        torch, dev, nq = self.job.torch, self.job.dev, self.nq if nq is None else nq
        return (torch.empty(nq, ef, dtype=torch.int32, device=dev), torch.empty(nq, ef, dtype=torch.float32, device=dev),
                torch.empty(nq, dtype=torch.int32, device=dev), torch.empty(nq, 3, dtype=torch.int32, device=dev))

    def run(self, outs, s=None, d_q=None):
        pid, dd, cnt, ctr = outs
        q = self.d_q if d_q is None else d_q
        if int(q.shape[0]) == 0:
            return                                           # more ranks than queries: this rank's block is empty
        self.hnsw.search_batch_device(s or self.search, q.data_ptr(), int(q.shape[0]), pid.data_ptr(), dd.data_ptr(), cnt.data_ptr(),
                                      ctr.data_ptr(), self.job.stream())


def phase_choose_ef(job, r, args, cfgd, k):
    """Exact ground truth for the head of this rank's block (scan with the same canonical distance), then the smallest ef_search
    of the ladder that reaches the recall target.  -> (chosen, recall, sweep, sample_out, gtq)"""
    hnsw, nq = r.hnsw, r.nq
    gt_req = cfgd["gtq"] if args.gt_queries < 0 else args.gt_queries
    gtq = min(gt_req or nq, nq)
    truth = hnsw.bruteforce(r.d_q[:gtq].cpu().numpy(), k)[0] if gtq else np.zeros((0, k), np.uint32)
    sweep, sample_out = {}, {}
    pq = min(args.parity_queries, nq)
    chosen, recall, last = None, 0.0, (args.ef or 100, 0.0)
    for ef in ([args.ef] if args.ef else [100, 200, 400, 800]):
        if chosen is not None and ef > 200:
            break                                   # 100 and 200 are always on the curve; beyond only while the target is missed
        hnsw.set_ef_search(ef)
        outs = r.alloc_out(ef)
        r.run(outs)
        job.sync()
        if nq:
            r.search.check_status()
        got = outs[0][:gtq, :k].cpu().numpy().astype(np.uint32)
        rec = float(np.mean([len(set(got[i].tolist()) & set(truth[i].tolist())) / k for i in range(gtq)])) if gtq else 1.0
        sweep[str(ef)] = round(rec, 4)
        sample_out[ef] = tuple(t[:pq].cpu().numpy() for t in outs)      # what the oracle is compared with below
        if chosen is None and (rec >= args.recall_target or args.ef):
            chosen, recall = ef, rec
        last = (ef, rec)
        del outs
    if chosen is None:
        chosen, recall = last
    return chosen, recall, sweep, sample_out, gtq


def phase_timed(job, r, outs, warmup, steps):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + device sync on both sides; the MAX over ranks."""
    for _ in range(warmup):
        r.run(outs)
    job.sync()
    if r.nq and getattr(r.search, "_ctx", None) is not None:
        r.search.filter_counts()                             # reset: what the reject filter does is counted over the K timed steps
    job.barrier()
    job.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.run(outs)
    job.sync()
    job.barrier()
    job.sync()
    elapsed = time.perf_counter() - t0
    elapsed = float(job.reduce(elapsed, job.torch.float64, "MAX"))
    if r.nq:
        r.search.check_status()
        r.filter_counts = r.search.filter_counts()           # (examined, rejected) summed over the K timed launches
    return elapsed


def answers_digest(arrays):
    """63 bits over ids + distance bits + counts + work counters (fits a signed int64 tensor)"""
    import hashlib
    h = hashlib.blake2b(digest_size=8)
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return int.from_bytes(h.digest(), "little") >> 1


def phase_replica_check(job, ida, hnsw, nq_total, dim, ef):
    """A bad broadcast must fail loudly.  Every rank answers the SAME sample (the head of the global batch) on its own replica;
    rank 0's answers are the yardstick (and are themselves held against the CPU oracle afterwards): one digest is broadcast and
    compared on every rank, and EVERY rank leaves with SystemExit when any replica disagrees.
    -> (replica_check object, d_sample, the sample's answers as numpy arrays)"""
    torch = job.torch
    rs_n = min(512, nq_total)
    d_sample = synth(torch, nq_total, dim, 123456790, job.dev)[:rs_n].contiguous()
    hnsw.set_ef_search(ef)                       # (the result buffers below are ef wide)
    r = Runner(job, ida, hnsw, d_sample)
    so = r.alloc_out(ef)
    r.run(so)
    job.sync()
    r.search.check_status()
    sample_np = [t.cpu().numpy() for t in so]
    mine = answers_digest(sample_np)
    ref = job.bcast_i64(mine, src=0)
    same = int(job.reduce(1 if ref == mine else 0, torch.int32, "MIN"))
    if not same:
        raise SystemExit(f"rank {job.rank}: replica answers differ from rank 0's on the {rs_n}-query sample (this rank's digest {mine:#x}, "
                         f"rank 0's {ref:#x}): a replicated index is not the built one")
    return {"queries": rs_n, "all_ranks_identical_to_rank0": True, "digest": f"{mine:#x}"}, d_sample, sample_np


def oracle_equals(o, pid, dd, cnt, ctr):
    return bool(np.array_equal(o.pid, pid.astype(np.uint32)) and np.array_equal(o.dist.view(np.uint32), dd.view(np.uint32))
                and np.array_equal(o.count, cnt.astype(np.uint32)) and np.array_equal(o.counters, ctr.astype(np.uint32)))


def wait_for_peers(pids, timeout=90.0):
    """Rank 0, after destroy_process_group: wait until the other ranks' processes are gone (their GPUs and HIP contexts with them)."""
    def running(pid):
        try:
            return open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0] != "Z"    # a zombie holds no GPU
        except OSError:
            return False

    t0 = time.time()
    alive = [p for p in pids if p != os.getpid()]
    while alive and time.time() - t0 < timeout:
        alive = [p for p in alive if running(p)]
        if alive:
            time.sleep(0.2)
    return not alive, round(time.time() - t0, 2)


def run_bench(job, args):
    """The whole measurement on one rank.  -> the output object on rank 0, None elsewhere."""
    torch, dist = job.torch, job.dist
    rank, world = job.rank, job.world
    cfgd = CONFIGS[args.config]

    import instant_distance_amd as ida
    from instant_distance_amd import dist as idd

    n, dim, k = args.n or cfgd["n"], args.dim or cfgd["dim"], args.k
    nq_cfg = args.nq or cfgd["nq"]
    split = cfgd["split"]
    nq_total = nq_cfg if split else nq_cfg * world
    builder = ida.Builder().max_batch(args.max_batch).device(job.local_rank)

    # ---- data + build (rank 0), replicate ----
    hnsw, d_pts, build = (None, None, {})
    if rank == 0:
        hnsw, d_pts, build = phase_build(job, ida, builder, n, dim)
    t_rep, replication, rep = 0.0, "single GPU", {}
    if job.multi:
        hnsw, t_rep, rep = phase_replicate(job, idd, hnsw, builder)
        replication = f"rank 0 built, {dist.get_backend()} broadcast of points/zero/upper device buffers"

    d_q, lo, hi = phase_queries(job, idd, nq_total, dim)
    nq = hi - lo
    r = Runner(job, ida, hnsw, d_q)
    search = r.search

    if args.traffic_child:
        # profiled child of measure_traffic(): the same index, the same queries; one calibration gather with a known byte count
        # (every row of the index once, in random order, through the 8-lanes-per-row loads of the walk), then full batches
        hnsw.set_ef_search(args.ef or 100)
        outs = r.alloc_out(args.ef or 100)
        perm = torch.randperm(n, device=job.dev).to(torch.int32).cpu().numpy().astype(np.uint32).reshape(1, n)
        hnsw.distances(d_q[:1].cpu().numpy(), perm)
        print("known_read_bytes_per_launch", n * hnsw.info().row_stride * 4 + n * 4, flush=True)
        # ... and the same for the compact rows of the reject filter (another request pattern: 320-B rows at C3)
        hnsw.filter_bounds(d_q[:1].cpu().numpy(), perm)
        print("known_compact_read_bytes_per_launch", n * ((hnsw.info().row_stride + 8 + 63) // 64 * 64) + n * 4, flush=True)
        for _ in range(2):
            r.run(outs)
        job.sync()
        search.check_status()
        return None

    # ---- ground truth + ef choice; every rank must time the same ef ----
    chosen, recall, sweep, sample_out, gtq = phase_choose_ef(job, r, args, cfgd, k)
    chosen = int(job.reduce(chosen, torch.int64, "MAX"))
    hnsw.set_ef_search(chosen)
    outs = r.alloc_out(chosen)

    # ---- timed region ----
    elapsed = phase_timed(job, r, outs, args.warmup, args.steps)

    replica_check, d_sample, sample_np = None, None, None
    if job.multi:
        replica_check, d_sample, sample_np = phase_replica_check(job, ida, hnsw, nq_total, dim, chosen)

    out = None
    if rank == 0:
        out = rank0_report(job, args, cfgd, ida, r, outs, hnsw, d_pts, build, dict(
            n=n, dim=dim, k=k, nq=nq, nq_total=nq_total, split=split, chosen=chosen, recall=recall, sweep=sweep, sample_out=sample_out,
            gtq=gtq, elapsed=elapsed, t_rep=t_rep, replication=replication, rep=rep, replica_check=replica_check, d_sample=d_sample,
            sample_np=sample_np))
    return out


def rank0_report(job, args, cfgd, ida, r, outs, hnsw, d_pts, build, m):
    """Rank 0 after the timed region: the JSON line's objects (roofline from the kernel's own counters and HIP events, the
    reference's scalar call pattern, the host-pointer rate, and the CPU oracle as checker / baseline)."""
    torch, world = job.torch, job.world
    n, dim, k, nq, nq_total, chosen = m["n"], m["dim"], m["k"], m["nq"], m["nq_total"], m["chosen"]
    search, d_q, run, alloc_out = r.search, r.d_q, r.run, r.alloc_out
    sample_out, replica_check = m["sample_out"], m["replica_check"]
    elapsed = m["elapsed"]
    ms_per_step = elapsed / args.steps * 1e3
    value = nq_total / (elapsed / args.steps)
    kt = search.kernel_times_ms(args.steps)
    ctr = outs[3].cpu().numpy().astype(np.int64)
    # SURVEY 8(d)'s per-query figure: every pushed candidate's f32 row (the reference fetches one per `push`, core/lib.rs:709-710)
    survey_bytes = int((ctr[:, 0] * 4 * dim + ctr[:, 1] * 256 + ctr[:, 2] * 128 + 8 * chosen).sum())
    # What THIS kernel's algorithm moves since round 6: wide batches look a candidate up in the one-byte-per-coordinate
    # copy of its row first (compact row: the stored row + 8 bytes, rounded to 64) and fetch the f32 row only of candidates that copy
    # cannot reject — push turns the others down without using their distance (core/lib.rs:712-714).  Counted by the kernel itself:
    # examined / rejected per launch (idist_search_ctx_filter_counts); a candidate pushed while `nearest` is not full skips the filter.
    info = hnsw.info()
    examined, rejected = [x / max(args.steps, 1) for x in getattr(r, "filter_counts", (0, 0))]
    compact_row = (info.row_stride + 8 + 63) // 64 * 64
    n_dist_launch = int(ctr[:, 0].sum())
    launch_bytes = int((n_dist_launch - rejected) * 4 * dim + examined * compact_row + (ctr[:, 1] * 256 + ctr[:, 2] * 128 + 8 * chosen).sum())
    kernel_ms = float(kt.mean()) if len(kt) else float("nan")
    achieved = launch_bytes / (kernel_ms * 1e-3) / 1e9
    # HBM bytes per launch: separate `rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE` child passes over this same command (PMC passes
    # cannot run inside the timed process); the newest committed result for the same workload is quoted beside it
    quoted, quoted_src, mall = None, None, None
    import glob
    for tp in sorted(glob.glob(os.path.join(ROOT, "profiles", "**", "traffic_r*.json"), recursive=True), key=os.path.basename, reverse=True):   # newest round first
        try:
            tj = json.load(open(tp))
            if tj.get("config", "C3") != args.config:
                continue
            quoted = tj.get("search_kernel_hbm_bytes_per_launch")
            mall = tj.get("mall")
            quoted_src = os.path.relpath(tp, ROOT) + " (separate PMC passes of this command, not this run)"
            break
        except Exception:  # noqa: BLE001
            continue
    traffic, traffic_why = (None, "skipped (--no-traffic, N > 1, or a 10M-point configuration)")
    if not args.no_traffic and world == 1 and job.cuda and n * dim <= 2_000_000_000:
        compact_b = examined * compact_row
        traffic, traffic_why = measure_traffic(args, n, dim, nq, chosen, mix={"rows": launch_bytes - compact_b, "compact": compact_b})
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic["bytes_per_launch"] if traffic else None,
                "traffic_over_algorithmic": round(traffic["bytes_per_launch"] / launch_bytes, 4) if traffic else None,
                "traffic_in_run": traffic if traffic else {"skipped": traffic_why},
                "traffic_quoted": quoted, "traffic_source": quoted_src,
                "mall_note": "FETCH_SIZE counts the L2's fabric-side requests: Infinity-Cache (MALL) hits are inside it, so "
                             "'traffic' is fabric bytes, an upper bound of DRAM bytes; that is how an algorithmic rate can "
                             "exceed the 6.29 TB/s streaming-copy rate of the HBM stacks", "mall": mall,
                "kernel": "search_kernel", "kernel_ms_avg": round(kernel_ms, 3),
                "alg_bytes_per_launch": launch_bytes, "alg_bytes_per_query": round(launch_bytes / max(nq, 1)),
                "alg_bytes_how": "f32 rows of the candidates the reject filter did not turn down + compact rows of the candidates it examined "
                                 "+ adjacency rows + results (kernel counters; = SURVEY 8(d)'s formula when nothing is filtered)",
                "reject_filter": {"examined_per_launch": round(examined), "rejected_per_launch": round(rejected),
                                  "rejected_share": round(rejected / examined, 4) if examined else None, "compact_row_bytes": compact_row,
                                  "pushes_per_launch": n_dist_launch},
                "survey_8d": {"bytes_per_launch": survey_bytes, "GBps": round(survey_bytes / (kernel_ms * 1e-3) / 1e9, 1),
                              "over_peak": round(survey_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                              "note": "SURVEY 8(d) books one f32 row per pushed candidate (n_dist*4D + n_exp0*256 + n_expU*128 + 8*ef): the rate a "
                                      "kernel that fetches them all would need to finish in this time — above the HBM peak means the filter "
                                      "went below those bytes, not that HBM ran faster"},
                "n_dist_per_query": round(float(ctr[:, 0].mean()), 1) if nq else 0.0, "n_exp0_per_query": round(float(ctr[:, 1].mean()), 1) if nq else 0.0,
                "n_expU_per_query": round(float(ctr[:, 2].mean()), 1) if nq else 0.0}

    # the reference's own call pattern: one query per call (`Hnsw::search`).  Outside the timed region.
    lat_n = min(32, nq)
    for i in range(lat_n):
        run(outs, d_q=d_q[i:i + 1])
    job.sync()
    single = {"gpu_kernel_ms_median": round(float(np.median(search.kernel_times_ms(lat_n))), 4) if lat_n else None, "queries": lat_n,
              "note": "nq = 1 per launch (the reference's scalar Hnsw::search); cpu = one oracle thread"}
    # the C ABI's host-pointer call: queries from host memory, results back to host memory (PCIe inclusive), never `value`
    q_host = d_q.cpu().numpy()
    hnsw.search_batch(q_host[:1], search)
    t0 = time.perf_counter()
    for i in range(lat_n):
        hnsw.search_batch(q_host[i:i + 1], search)
    single["gpu_wall_ms_host_pointers"] = round((time.perf_counter() - t0) / max(lat_n, 1) * 1e3, 4)
    # T host threads, one Search each, scalar calls on one shared index (core/lib.rs:352-356)
    thr = {}
    for T in [int(x) for x in args.threads.split(",") if x]:
        thr[str(T)] = {"gpu_calls_per_s": round(scalar_calls(hnsw, ida, q_host, T, max(200, 1600 // T)), 1),
                       "gpu_kernel_ms_mean": None if scalar_calls.kernel_ms is None else round(scalar_calls.kernel_ms, 4)}
    single["threads"] = thr
    hnsw.search_batch(q_host, search)
    t0 = time.perf_counter()
    for _ in range(3):
        res_host = hnsw.search_batch(q_host, search, counters=True)
    pcie = {"qps": round(nq * 3 / (time.perf_counter() - t0), 1),
            "note": f"idist_search_batch with host pointers: {nq * dim * 4 >> 20} MB of queries in, {nq * chosen * 8 >> 20} MB of results out per call, pageable memory"}
    run(outs)                      # restore the full-batch outputs the checks below read
    job.sync()
    assert np.array_equal(res_host.pid, outs[0].cpu().numpy().astype(np.uint32))
    del res_host

    checks = None
    if args.check:
        pid_t, dd_t, cnt_t = outs[0], outs[1], outs[2]
        srt = torch.sort(pid_t, dim=1).values
        outs2 = alloc_out(chosen)
        run(outs2, ida.Search())                                     # a fresh Search: same answers (idempotence)
        job.sync()
        ns = min(64, n)
        self_q = d_pts[:ns].cpu().numpy()
        sres = hnsw.search_batch(self_q, ida.Search())                # narrow batch: four waves per query
        checks = {"count_is_ef": bool((cnt_t == min(chosen, n)).all().item()),
                  "sorted_nearest_first": bool((dd_t[:, :-1] <= dd_t[:, 1:]).all().item()),
                  "ids_unique_per_query": bool((srt[:, 1:] != srt[:, :-1]).all().item()),
                  "idempotent": bool(torch.equal(outs2[0], pid_t) and torch.equal(outs2[1], dd_t)),
                  "self_query_first_at_distance_0": bool(np.array_equal(sres.pid[:, 0], np.arange(ns)) and np.all(sres.distance[:, 0] == 0))}
        del outs2, srt

    cpu, parity = None, None
    need_gb = (n * dim * 4 + n * 256 * 2) / 2**30 + 4
    try:
        avail_gb = int(next(l for l in open("/proc/meminfo") if l.startswith("MemAvailable")).split()[1]) / 2**20
    except Exception:  # noqa: BLE001
        avail_gb = 1e9
    oix = None
    if not args.no_cpu_baseline and avail_gb < need_gb:
        parity = {"skipped": f"the oracle needs the points on the host: {need_gb:.0f} GB, {avail_gb:.0f} GB available"}
    elif not args.no_cpu_baseline:
        from oracle import pyoracle as po
        zero, layers = hnsw.into_parts()
        pts_h = d_pts.cpu().numpy()
        # (the oracle reads the arrays in place: a second host copy of 10M x 768 points would be another 31 GB)
        oix = po.Index.from_arrays(pts_h, zero, layers, po.default_config(ef_search=chosen), borrow=True)
        cores = effective_cores()
    if oix is not None and replica_check is not None:
        # the yardstick of the replica check is itself held against the oracle: rank 0's answers for the sample == the oracle's on
        # the exported graph
        o = oix.search(m["d_sample"].cpu().numpy(), threads=cores)
        ok = oracle_equals(o, *m["sample_np"])
        replica_check["rank0_identical_to_oracle"] = ok
        if not ok:
            raise SystemExit("rank 0: GPU answers for the replica-check sample differ from the CPU oracle's on the exported graph")
        if world > 1:
            parity = {"queries": replica_check["queries"], f"ef{chosen}": ok, "all_identical": ok,
                      "note": "N > 1: rank 0's answers for the replica-check sample against the oracle on the exported graph"}
    if oix is not None and world == 1:
        # (CPU timing on rank 0 at N = 1 only)
        if checks is not None:
            valid = zero != 0xFFFFFFFF
            checks["layer_sizes_match_reference"] = [l.shape[0] for l in layers] == po.layer_sizes(n)[1:]
            checks["rows_prefix_valid"] = bool(np.all(valid[:, :-1] >= valid[:, 1:]))
            checks["row_ids_in_range"] = bool(np.all(zero[valid] < n))
            checks["min_degree"] = int(valid.sum(1).min())
            checks["no_self_links"] = bool(not np.any(zero == np.arange(n, dtype=np.uint32)[:, None]))
            del valid
        q_h = q_host
        pq = min(args.parity_queries, nq)
        # ---- parity: ids / order / counts / distance bits / work counters of a query sample, per ef ----
        parity = {"queries": pq}
        for ef, got in sorted(sample_out.items()):
            oix.set_ef_search(ef)
            parity[f"ef{ef}"] = oracle_equals(oix.search(q_h[:pq], threads=cores), *got)
        parity["all_identical"] = all(v for k_, v in parity.items() if k_.startswith("ef"))
        oix.set_ef_search(chosen)
        # bounded sample: ~10-30 core-seconds of CPU work; best of 3 passes (thread start-up noise)
        probe = min(nq, 8 * cores)
        t0 = time.perf_counter(); oix.search(q_h[:probe], threads=cores); tp_ = time.perf_counter() - t0
        core_s_per_q = tp_ * min(cores, probe) / probe
        sample = args.cpu_sample or int(min(nq, max(probe, 20.0 / max(core_s_per_q, 1e-6))))
        tc = 1e30
        for _ in range(3):
            t0 = time.perf_counter(); ores = oix.search(q_h[:sample], threads=cores); tc = min(tc, time.perf_counter() - t0)
        same = bool(np.array_equal(ores.pid, outs[0][:sample].cpu().numpy().astype(np.uint32)))
        same_all = oracle_equals(ores, *[t[:sample].cpu().numpy() for t in outs])
        t0 = time.perf_counter(); oix.search(q_h[:200], threads=1)
        single["cpu_ms_per_query_one_thread"] = round((time.perf_counter() - t0) / min(200, nq) * 1e3, 4)
        for T, row in thr.items():
            nn = min(nq, max(200, 100 * int(T)))
            t0 = time.perf_counter(); oix.search(q_h[:nn], threads=int(T))
            row["cpu_oracle_calls_per_s"] = round(nn / (time.perf_counter() - t0), 1)
        # build baseline: the oracle's threaded build (per-layer parallel-for + per-node locks, the rayon path of
        # core/lib.rs:316-318) on a PREFIX of the same points — the rate falls with n, so this flatters the CPU
        nb_ = min(n, cfgd["cpu_build"] if args.cpu_build_sample < 0 else args.cpu_build_sample)
        if nb_:
            t0 = time.perf_counter(); po.Index.build(pts_h[:nb_], po.default_config(), threads=cores); tb_ = time.perf_counter() - t0
            build["cpu_baseline"] = {"value": round(nb_ / tb_, 1), "unit": "points/s", "cores": cores, "kind": "port",
                                     "sample": f"first {nb_} of the {n} points, {cores} threads (prefix: optimistic for the CPU)",
                                     "seconds": round(tb_, 2)}
            build["gpu_over_cpu"] = round(build["points_per_s"] / build["cpu_baseline"]["value"], 1)
        # the FULL-size CPU build takes minutes (1M x 300: 186 s on 16 threads), so it is not repeated in every run: the newest committed
        # measurement of it for this shape is quoted beside the prefix (scripts/probe_build_quality.py: same points, same oracle, recall
        # of both graphs through the engine)
        for qp in sorted(glob.glob(os.path.join(ROOT, "profiles", "**", "probe_r*_build_quality_*.jsonl"), recursive=True), key=os.path.basename, reverse=True):
            try:
                rows = [json.loads(l) for l in open(qp) if l.startswith("{")]
                full = [x for x in rows if x.get("builder", "").startswith("cpu oracle") and x.get("n") == n and x.get("dim") == dim]
                if full:
                    build["cpu_baseline_full_size_quoted"] = {"value": round(n / full[-1]["build_seconds"], 1), "unit": "points/s", "seconds": full[-1]["build_seconds"],
                                                              "builder": full[-1]["builder"], "recall_at_10_ef100": full[-1].get("recall_at_10_ef100"),
                                                              "source": os.path.relpath(qp, ROOT) + " (not this run)"}
                    break
            except Exception:  # noqa: BLE001
                continue
        cpu = {"value": round(sample / tc, 1), "unit": "queries/s", "cores": cores, "kind": "port",
               "sample": f"first {sample} of the {nq} queries, same graph, ef_search={chosen}, {cores} threads = the "
                         f"container's CPU quota ({os.cpu_count()} logical CPUs visible) "
                         "(C oracle = restated reference, not the Rust crate)",
               "seconds": round(tc, 2), "ids_identical_to_gpu": same,
               "ids_distance_bits_counts_and_work_counters_identical_to_gpu": same_all}

    recall = m["recall"]
    out = {"commit": source_stamp(),
           "metric": f"queries/sec @ recall@10>=0.95, {'1M' if n == 1_000_000 else n}x{dim}-d f32; index build points/sec",
           "value": round(value, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if m["split"] else "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{cfgd['what']} — {n}x{dim}-d f32 fastText-shape synthetic (32-d latent, L2-normalised), "
                                  f"Builder::build on GPU + {nq}-query Hnsw::search batch per GPU, k={k}",
                      "name": args.config, "n": n, "dim": dim, "queries_per_gpu": nq, "k": k, "ef_search": chosen,
                      "recall_at_10": round(recall, 4), "recall_target": args.recall_target,
                      "recall_target_met": bool(recall >= args.recall_target), "recall_queries": m["gtq"],
                      "ef_sweep_recall": m["sweep"], "parallelism": f"query-shard x{world}, index replicated",
                      "replication": m["replication"], "replicate_seconds": round(m["t_rep"], 3), **m["rep"]},
           "build": build, "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "single_query": single,
           "pcie_inclusive": pcie}
    try:                                                     # rank 0's footprint (C5: 30.7 GB of points + the index + the oracle's leg)
        import resource
        free_b, total_b = torch.cuda.mem_get_info() if job.cuda else (0, 0)
        out["memory"] = {"host_rss_peak_gb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0, 2),
                         "hbm_in_use_at_end_gb": round((total_b - free_b) / 1e9, 2), "hbm_total_gb": round(total_b / 1e9, 1)}
    except Exception:  # noqa: BLE001
        pass
    if replica_check is not None:
        out["replica_check"] = replica_check
    if checks is not None:
        out["checks"] = checks
    if cpu:
        out["gpu_over_cpu"] = round(value / cpu["value"], 2)
    return out


def inproc_rccl_child(args, world, n, dim, chosen):
    """The C ABI's own replication (idist_replicate_rccl: ONE process, ncclCommInitAll over the devices, one grouped ncclBroadcast
    per buffer) next to the torch.distributed path.  A child process: whatever RCCL does there (a hang, a crash) costs the line
    nothing but this entry."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--rccl-child", str(world), "--config", args.config, "--n", str(n),
                            "--dim", str(dim), "--ef", str(chosen), "--max-batch", str(args.max_batch)], capture_output=True, text=True, timeout=420)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if lines else {"error": (r.stderr or r.stdout)[-300:]}
    except Exception as e:  # noqa: BLE001
        return {"skipped": repr(e)[:200]}


def self_launch_command(argv, gpus, port):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: the command main() starts itself under — one rank per GPU
    over RCCL, the driver's own shape (torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def error_line(args, msg):
    """ONE JSON line for a run that cannot start (rank 0's line has the same leading keys)."""
    return json.dumps({"metric": "queries/sec @ recall@10>=0.95, 1Mx300-d f32; index build points/sec", "value": None, "unit": "queries/s",
                       "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "error": msg[-600:]})


def self_launch(args, argv):
    """--gpus N > 1 without WORLD_SIZE / RANK in the environment.  Fails fast with one JSON line when the node has fewer than N
    devices; otherwise runs the N ranks as a child (bounded), passes rank 0's line through, and prints an error line if none came."""
    import socket
    import subprocess

    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        print(error_line(args, f"--gpus {args.gpus} but {have} MI355X device(s) visible on this node"), flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = self_launch_command(argv, args.gpus, port)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=args.launch_timeout)
        rc, text = r.returncode, r.stdout
    except subprocess.TimeoutExpired as e:
        rc, text = 124, (e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or ""))
    lines = [l for l in text.splitlines() if l.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
        return 0 if rc == 0 else rc
    print(error_line(args, f"torch.distributed.run exited with {rc} and no result line: " + text[-400:]), flush=True)
    return rc or 1


def main():
    args = parse_args()
    cfgd = CONFIGS[args.config]
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and not args.rccl_child:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    import torch

    if args.rccl_child:
        return rccl_child(args, cfgd, torch)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(error_line(args, f"--gpus {args.gpus} under a launcher with WORLD_SIZE={world}"), flush=True)
        raise SystemExit(2)
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        if rank == 0:
            print(error_line(args, "bench.py needs one MI355X per rank: instant_distance_amd has no CPU path"), flush=True)
        raise SystemExit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # Launched by torch.distributed.run (RANK + MASTER_ADDR in the environment): one rank per GPU over RCCL — also at
    # --nproc-per-node 1, where the same replicate / agree / barrier / digest lines run with a world of one.
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        import datetime

        import torch.distributed as dist
        # every collective wait is bounded: a rank that died leaves the others with an error, not a hang
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=args.collective_timeout))
    job = Job(torch, rank, world, local_rank, dev, dist)
    out = run_bench(job, args)
    if args.traffic_child:
        return
    pids = None
    if job.multi:
        pids = job.gather_i64(os.getpid())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if job.multi and not args.no_inproc_rccl:
            try:
                gone, waited = wait_for_peers(pids)
                child = inproc_rccl_child(args, world, out["config"]["n"], out["config"]["dim"], out["config"]["ef_search"])
                child["peer_ranks_exited_before_launch"] = gone
                child["waited_for_peers_s"] = waited
            except Exception as e:  # noqa: BLE001 — the line is printed whatever the in-process RCCL child does
                child = {"error": repr(e)[:300]}
            out["config"]["replicate_rccl_in_process"] = child
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
