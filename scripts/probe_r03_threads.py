"""The reference's concurrency model on the GPU (core/lib.rs:352-356): T host threads, one Search each, scalar
idist_search_batch(nq = 1) calls on ONE shared C3 index — aggregate calls/s under runtime settings that decide whether
the one-workgroup kernels of different streams overlap: hardware queues (GPU_MAX_HW_QUEUES), the timing events around
each launch.  Every variant runs in its own process (the settings are read when the HIP runtime starts).
usage: python scripts/probe_r03_threads.py out.jsonl [variant,variant,...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if os.environ.get("PB_CHILD"):
    import numpy as np
    import torch

    import bench
    import instant_distance_amd as ida

    dev = torch.device("cuda", 0)
    n, dim = 1_000_000, 300
    d_pts = bench.synth(torch, n, dim, 123456789, dev)
    q = bench.synth(torch, 4096, dim, 123456790, dev).cpu().numpy()
    torch.cuda.synchronize()
    h = ida.Hnsw.from_device_points(d_pts.data_ptr(), n, dim, ida.Builder())
    row = {"variant": os.environ["PB_CHILD"]}
    for T in (1, 2, 4, 8, 16, 32, 64):
        row[f"T{T}"] = round(bench.scalar_calls(h, ida, q, T, max(100, 2400 // T)), 1)
        row[f"T{T}_kernel_ms"] = round(bench.scalar_calls.kernel_ms, 4)
    print("ROW " + json.dumps(row), flush=True)
    sys.exit(0)

fo = open(sys.argv[1], "a")
VARIANTS = [("default", {}), ("no_combine", {"IDIST_COMBINE": "0"}), ("sync_stream", {"IDIST_SYNC": "stream"}), ("hwq4", {"GPU_MAX_HW_QUEUES": "4"}), ("hwq8", {"GPU_MAX_HW_QUEUES": "8"}), ("hwq16", {"GPU_MAX_HW_QUEUES": "16"}),
            ("no_events", {"IDIST_KERNEL_EVENTS": "0"}), ("no_events_hwq16", {"IDIST_KERNEL_EVENTS": "0", "GPU_MAX_HW_QUEUES": "16"}),
            ("no_events_hwq32_staged", {"IDIST_KERNEL_EVENTS": "0", "GPU_MAX_HW_QUEUES": "32", "IDIST_NO_ZERO_COPY": "1"})]
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for name, env in VARIANTS:
    if only and name not in only:
        continue
    e = dict(os.environ, PB_CHILD=name, **env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=e, capture_output=True, text=True, timeout=600)
    rows = [l[4:] for l in r.stdout.splitlines() if l.startswith("ROW ")]
    line = rows[-1] if rows else json.dumps({"variant": name, "err": (r.stderr or r.stdout)[-400:]})
    print(line, flush=True)
    fo.write(line + "\n")
    fo.flush()
