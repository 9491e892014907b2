"""Condense rocprofv3 outputs of scripts/profile_bench.sh into one JSON (for profiles/)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
res = {}


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


def short(name):
    for k in ("search_kernel", "build_insert_kernel", "build_update_kernel", "bruteforce_kernel", "distance_batch_kernel",
              "permute_rows_kernel", "snapshot_kernel", "validate_rows_kernel"):
        if k in name:
            return k
    return name[:60]


# kernel stats
for f in find("trace", "*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    res["kernel_stats"] = [{k: r[k] for k in r} for r in rows[:12]]
# per-kernel average from the trace itself
for f in find("trace", "*kernel_trace.csv"):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        name = short(r.get("Kernel_Name", ""))
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        agg[name][0] += 1
        agg[name][1] += dur
    res["kernel_trace_avg_ms"] = {k: {"calls": v[0], "total_ms": round(v[1], 3), "avg_ms": round(v[1] / v[0], 4)} for k, v in agg.items()}


def pmc(sub, counter):
    agg = defaultdict(list)
    for f in find(sub, "*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                agg[short(r.get("Kernel_Name", ""))].append(float(r["Counter_Value"]))
    return {k: {"dispatches": len(v), "mean": sum(v) / len(v), "last": v[-1], "max": max(v)} for k, v in agg.items()}


res["pmc_FETCH_SIZE_KB"] = pmc("pmc_fetch", "FETCH_SIZE")
res["pmc_WRITE_SIZE_KB"] = pmc("pmc_write", "WRITE_SIZE")
res["calib_FETCH_SIZE_KB"] = pmc("calib_fetch", "FETCH_SIZE")
try:
    known = None
    for line in open(os.path.join(out, "calib_fetch.log")):
        if line.startswith("known_read_bytes_per_launch"):
            known = int(line.split()[1])
    c = res["calib_FETCH_SIZE_KB"].get("distance_batch_kernel")
    if known and c:
        res["calibration"] = {"known_read_bytes": known, "reported_bytes": c["last"] * 1024,
                              "correction_factor": known / (c["last"] * 1024)}
except Exception as e:  # noqa: BLE001
    res["calibration_error"] = str(e)
try:
    res["bench"] = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    res["bench_error"] = str(e)
print(json.dumps(res, indent=1))
