"""Where a build step's time goes along its dependency chain  descents(k) -> selection(k) -> carry-over -> memoised updates(k) ->
full re-selections(k) -> descents(k + 2)  from a rocprofv3 --kernel-trace db: per class of steps (by the descent's grid), the mean
kernel durations and the mean GAPS between one kernel's end and its successor's start (dependent-launch latency), and the mean
step period (descent start to descent start).
usage: trace_chain.py <dir> [out.json]"""
import glob
import json
import sqlite3
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]
cur = sqlite3.connect(f).cursor()
rows = cur.execute("select name, grid_x, workgroup_x, start, end from kernels where name like '%idist::%' order by start").fetchall()
seq = defaultdict(list)
for name, grid, wg, s, e in rows:
    k = name.split("idist::")[1].split("<")[0].split("(")[0]
    seq[k].append((s, e, grid // max(wg, 1), wg))
ins = seq["build_insert_kernel"]
n = len(ins)
sel = (seq.get("build_select_mfma_kernel") or seq.get("build_select_kernel") or [])
fast, upd, cp = seq.get("build_update_fast_kernel", []), seq.get("build_update_kernel", []), seq.get("copy_rows_kernel", [])
print("steps", n, "select", len(sel), "fast", len(fast), "update", len(upd), "copy", len(cp))
t0 = ins[0][0]
span = (max(r[4] for r in rows if "build" in r[0] or "copy_rows" in r[0]) - t0) / 1e6
classes = defaultdict(lambda: defaultdict(float))
for i in range(n):
    if i >= len(sel) or i >= len(fast) or i >= len(upd):
        break
    wgs, wg = ins[i][2], ins[i][3]
    c = ("quad" if wg == 256 else "wave") + (f" <= {1 << max(0, (wgs - 1)).bit_length()}" if wgs <= 512 else " wide")
    d = classes[c]
    d["steps"] += 1
    d["A_ms"] += (ins[i][1] - ins[i][0]) / 1e6
    d["gap_A_to_A2_ms"] += (sel[i][0] - ins[i][1]) / 1e6
    d["A2_ms"] += (sel[i][1] - sel[i][0]) / 1e6
    d["gap_A2_to_F_ms"] += (fast[i][0] - sel[i][1]) / 1e6
    d["F_ms"] += (fast[i][1] - fast[i][0]) / 1e6
    d["gap_F_to_B2_ms"] += (upd[i][0] - fast[i][1]) / 1e6
    d["B2_ms"] += (upd[i][1] - upd[i][0]) / 1e6
    if i + 2 < n:
        d["gap_B2_to_A_of_k_plus_2_ms"] += (ins[i + 2][0] - upd[i][1]) / 1e6
        d["chain_ms"] += (ins[i + 2][0] - ins[i][0]) / 1e6
    if i + 1 < n:
        d["period_ms"] += (ins[i + 1][0] - ins[i][0]) / 1e6
out = {"span_ms": round(span, 3), "steps": n, "classes": {}}
for c, d in sorted(classes.items(), key=lambda kv: kv[0]):
    s = d.pop("steps")
    out["classes"][c] = {"steps": int(s), "total_period_ms": round(d["period_ms"], 2), **{k: round(v / s, 4) for k, v in d.items()}}
    print(c, json.dumps(out["classes"][c]))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
