"""Condense the rocprofv3 (rocpd SQLite) outputs of scripts/profile_bench.sh into one JSON
+ a markdown summary for profiles/.  Usage: summarize_profile.py gpurun_out/<tag> [profiles/<name>]"""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

out = sys.argv[1]
dest = sys.argv[2] if len(sys.argv) > 2 else None
what = sys.argv[3] if len(sys.argv) > 3 else "`python bench.py` (C3: 1M x 300-d, 10k queries, ef_search=100)"
# `summarize_profile.py <dir> <dest> [what] --from-json <summary.json>`: re-render the .md / .json / traffic files from a summary this
# script printed on the GPU box (the raw databases do not travel back)
FROM_JSON = sys.argv[sys.argv.index("--from-json") + 1] if "--from-json" in sys.argv else None
if FROM_JSON and what == "--from-json":
    what = "`python bench.py` (C3: 1M x 300-d, 10k queries, ef_search=100)"
res = {}
try:   # the commit the profiled tree was built from (scripts/stamp.py; .git does not travel to the GPU box)
    res["commit"] = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".build_commit")).read().strip()
except OSError:
    res["commit"] = None
KEYS = ("search_kernel", "build_insert_kernel", "build_select_mfma_kernel", "copy_rows_kernel", "build_select_kernel", "build_update_fast_kernel", "build_update_simple_kernel", "build_update_kernel",
        "bruteforce_kernel", "distance_batch_kernel", "mfma_dist_kernel", "rerank_kernel", "kth_threshold_kernel",
        "row_norms_kernel", "permute_rows_kernel", "snapshot_kernel", "validate_rows_kernel", "filter_rows_kernel", "filter_bound_kernel")


def short(name):
    for k in KEYS:
        if k in name:
            return k
    return None


def dbs(sub):
    return sorted(glob.glob(os.path.join(out, sub, "**", "*_results.db"), recursive=True))


if FROM_JSON:
    res = json.load(open(FROM_JSON))
else:
    for f in dbs("trace"):
        cur = sqlite3.connect(f).cursor()
        rows = cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
        res["kernel_stats"] = [{"kernel": r[0][:90], "calls": r[1], "total_us": round(r[2], 1), "avg_us": round(r[3], 1),
                                "pct": round(r[4], 3)} for r in rows[:8]]
        rows = cur.execute("select name, vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, grid_x from kernels "
                           "group by name").fetchall()
        res["kernel_resources"] = {short(r[0]): {"vgpr": r[1], "sgpr": r[2], "lds_bytes": r[3], "scratch": r[4],
                                                  "workgroup": r[5], "grid_threads_example": r[6]} for r in rows if short(r[0])}


    def pmc(sub, counter):
        agg = defaultdict(list)
        for f in dbs(sub):
            cur = sqlite3.connect(f).cursor()
            for k, c, v, d in cur.execute("select kernel_name, counter_name, value, duration from counters_collection"):
                if c == counter and short(k):
                    agg[short(k)].append((v, d))
        return {k: {"dispatches": len(v), "mean_KB": round(sum(x[0] for x in v) / len(v), 1), "last_KB": round(v[-1][0], 1),
                    "sum_KB": round(sum(x[0] for x in v), 1), "mean_ms": round(sum(x[1] for x in v) / len(v) / 1e6, 3)} for k, v in agg.items()}


    res["pmc_FETCH_SIZE"] = pmc("pmc_fetch", "FETCH_SIZE")
    res["pmc_WRITE_SIZE"] = pmc("pmc_write", "WRITE_SIZE")
    res["calib_FETCH_SIZE"] = pmc("calib_fetch", "FETCH_SIZE")
    factor = None
    try:
        known = None
        for line in open(os.path.join(out, "calib_fetch.log")):
            if line.startswith("known_read_bytes_per_launch"):
                known = int(line.split()[1])
        c = res["calib_FETCH_SIZE"].get("distance_batch_kernel")
        if known and c:
            factor = known / (c["last_KB"] * 1024)
            res["calibration"] = {"kernel": "distance_batch_kernel over a random permutation of all 1M rows (8 lanes x float4 per row)",
                                  "known_read_bytes": known, "reported_bytes": c["last_KB"] * 1024,
                                  "correction_factor": round(factor, 4),
                                  "note": "gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md §HBM)"}
    except Exception as e:  # noqa: BLE001
        res["calibration_error"] = str(e)
    try:
        res["bench"] = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        res["bench_error"] = str(e)
    # the bench line printed by the TRACED run itself: its HIP-event kernel time must agree with the trace
    try:
        for line in open(os.path.join(out, "trace.log")):
            if line.startswith("{") and '"metric"' in line:
                res["bench_traced_run"] = json.loads(line)
        for f in dbs("trace"):
            cur = sqlite3.connect(f).cursor()
            # full-batch launches only (the ef sweep's other widths, the parity sample and the single-query probes use smaller grids /
            # other walks)
            d = [r[0] for r in cur.execute("select end - start from kernels where name like '%search_kernel%' and grid_x = "
                                           "(select max(grid_x) from kernels where name like '%search_kernel%') order by start")]
            res["search_kernel_full_batch_launches"] = {"calls": len(d), "avg_us": round(sum(d) / max(len(d), 1) / 1e3, 1)}
            steps = res.get("bench_traced_run", {}).get("steps", 5)
            if len(d) >= steps + 1:
                timed = d[-(steps + 1):-1]      # the last launch restores the outputs after the single-query probes
                res["agreement"] = {"rocprof_avg_ms_of_the_timed_launches": round(sum(timed) / len(timed) / 1e6, 3),
                                    "bench_hip_event_avg_ms_same_run": res["bench_traced_run"]["roofline"]["kernel_ms_avg"],
                                    "timed_launches": len(timed)}
    except Exception as e:  # noqa: BLE001
        res["agreement_error"] = str(e)
    # L2 (TCC) view of the search kernel's full-batch launches: hit rate, reads that left the L2 for the fabric (EA), and how many
    # of those were addressed to the DRAM controllers.  The Infinity Cache (MALL) sits BEHIND the fabric port: its hits are part
    # of both EA counters, rocprofv3 on this build lists no MALL counter, so "fabric bytes" is an upper bound of DRAM bytes.
    tcc = defaultdict(lambda: defaultdict(list))
    for f in dbs("pmc_tcc"):
        cur = sqlite3.connect(f).cursor()
        for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
            if short(k):
                tcc[short(k)][c].append(v)
    if tcc.get("search_kernel"):
        t = {c: v[-1] for c, v in tcc["search_kernel"].items()}           # the last launch is a full 10k-query batch
        hit, miss = t.get("TCC_HIT_sum", 0.0), t.get("TCC_MISS_sum", 0.0)
        rd, rdd = t.get("TCC_EA0_RDREQ_sum", 0.0), t.get("TCC_EA0_RDREQ_DRAM_sum", 0.0)
        res["l2_and_fabric"] = {"kernel": "search_kernel (last full-batch launch)", "TCC_HIT": hit, "TCC_MISS": miss,
                                "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None,
                                "EA_read_requests": rd, "EA_read_requests_to_DRAM_controllers": rdd,
                                "share_of_fabric_reads_addressed_to_DRAM": round(rdd / rd, 4) if rd else None,
                                "mall_note": "Infinity-Cache hits are inside both EA counters (the MALL sits behind the fabric port) and "
                                             "rocprofv3 lists no MALL counter on gfx950: FETCH_SIZE is fabric bytes, an upper bound of DRAM bytes"}
    sf, sw = res["pmc_FETCH_SIZE"].get("search_kernel"), res["pmc_WRITE_SIZE"].get("search_kernel")
    if sf and sw and factor:
        fetch = sf["last_KB"] * 1024 * factor
        write = sw["last_KB"] * 1024
        res["traffic"] = {"config": res.get("bench", {}).get("config", {}).get("name", "C3"), "mall": res.get("l2_and_fabric"),
                          "search_kernel_fetch_bytes_raw": sf["last_KB"] * 1024, "search_kernel_fetch_bytes_corrected": round(fetch),
                          "search_kernel_write_bytes": round(write), "search_kernel_hbm_bytes_per_launch": round(fetch + write)}
        if "bench" in res:
            alg = res["bench"]["roofline"]["alg_bytes_per_launch"]
            res["traffic"]["alg_bytes_per_launch"] = alg
            res["traffic"]["traffic_over_algorithmic"] = round((fetch + write) / alg, 3)
    
# Round 6: the filtered walk mixes 320-B compact rows (x 1.674) with f32 rows (x 1.898); this session's separate PMC passes carry the
# f32-row calibration only, so `traffic_over_algorithmic` above is an UPPER bound.  The bench line of the same session calibrated both
# patterns in its own PMC passes and weighted them by the launch's byte mix: apply that factor to this session's raw FETCH_SIZE too.
if isinstance(res.get("traffic"), dict) and "bench" in res:
    tir = res["bench"]["roofline"].get("traffic_in_run") or {}
    if tir.get("fetch_correction_factor_compact_rows") and "search_kernel_fetch_bytes_raw" in res["traffic"]:
        t = res["traffic"]
        mixed = t["search_kernel_fetch_bytes_raw"] * tir["fetch_correction_factor"] + t["search_kernel_write_bytes"]
        t["fetch_correction_factor_bench_line_both_patterns"] = tir["fetch_correction_factor"]
        t["search_kernel_hbm_bytes_per_launch_both_patterns"] = round(mixed)
        t["traffic_over_algorithmic_both_patterns"] = round(mixed / t["alg_bytes_per_launch"], 3)
        t["note"] = ("search_kernel_hbm_bytes_per_launch applies the f32-row factor to every request (upper bound); *_both_patterns uses "
                     "the bench line's factor (f32 rows x 1.898, compact rows x 1.674, weighted by the launch's byte mix)")

print(json.dumps(res, indent=1))
if dest:
    json.dump(res, open(dest + ".json", "w"), indent=1)
    if "traffic" in res:
        res["traffic"]["commit"] = res["commit"]
        json.dump(res["traffic"], open(os.path.join(os.path.dirname(dest), "traffic_" + os.path.basename(dest).split("_")[-1] + ".json"), "w"), indent=1)
    if isinstance(res.get("traffic"), dict):
        res["traffic"]["commit"] = res["commit"]
    with open(dest + ".md", "w") as f:
        f.write(f"# rocprofv3 summary ({os.path.basename(dest)})\n\nCommit: `{res['commit']}`.  Command: {what} under "
                "`rocprofv3 --kernel-trace --stats`" + (", then separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` / TCC passes" if res["pmc_FETCH_SIZE"] else "") + ".\n\n")
        f.write("## kernel-trace --stats (top kernels)\n\n| kernel | calls | total µs | avg µs | % |\n|---|---|---|---|---|\n")
        for r in res.get("kernel_stats", []):
            f.write(f"| `{r['kernel'][:70]}` | {r['calls']} | {r['total_us']} | {r['avg_us']} | {r['pct']} |\n")
        if "search_kernel_full_batch_launches" in res:
            f.write("\n`search_kernel` full-batch launches only (the table above also counts the launches of the ef sweep at ef 200, whose "
                    "kernel runs longer, and the one-query probes): " + json.dumps(res["search_kernel_full_batch_launches"]) + "\n")
        f.write("\n## resources\n\n```\n" + json.dumps(res.get("kernel_resources", {}), indent=1) + "\n```\n")
        f.write("\n## PMC (per dispatch, KB as reported)\n\n```\nFETCH_SIZE " + json.dumps(res["pmc_FETCH_SIZE"], indent=1) +
                "\nWRITE_SIZE " + json.dumps(res["pmc_WRITE_SIZE"], indent=1) + "\n```\n")
        f.write("\n## FETCH_SIZE calibration on the gather pattern\n\n```\n" + json.dumps(res.get("calibration", {}), indent=1) + "\n```\n")
        f.write("\n## search_kernel fabric traffic per launch (FETCH_SIZE counts Infinity-Cache hits too)\n\n```\n" + json.dumps(res.get("traffic", {}), indent=1) + "\n```\n")
        if "l2_and_fabric" in res:
            f.write("\n## L2 hit rate and fabric reads of the search kernel\n\n```\n" + json.dumps(res["l2_and_fabric"], indent=1) + "\n```\n")
        if "agreement" in res:
            f.write("\n## search_kernel duration: rocprofv3 trace vs bench.py's own HIP events (the traced run)\n\n```\n" +
                    json.dumps(res["agreement"], indent=1) + "\n```\n(clock state moves the kernel time by several % between "
                    "runs on one box; compare numbers of the same run)\n")
        if "bench" in res:
            f.write("\n## bench line of the same session (un-profiled run)\n\n```json\n" + json.dumps(res["bench"]) + "\n```\n")
        if "bench_traced_run" in res:
            f.write("\n## bench line printed by the traced run\n\n```json\n" + json.dumps(res["bench_traced_run"]) + "\n```\n")
