"""Per-dispatch durations of the build kernels from a rocprofv3 --kernel-trace output, bucketed by the step's grid size.
usage: trace_steps.py <dir>"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)):
    cur = sqlite3.connect(f).cursor()
    rows = cur.execute("select name, grid_x, end - start, start from kernels where name like '%idist::build%' order by start").fetchall()
    agg = defaultdict(lambda: [0, 0.0])
    for name, grid, dur, _ in rows:
        k = name.split("idist::")[1].split("<")[0].split("(")[0]
        b = "grid<=16K" if grid <= 16384 else ("grid<=64K" if grid < 65536 else "full")
        agg[(k, b)][0] += 1
        agg[(k, b)][1] += dur / 1e6
    for (k, b), (c, ms) in sorted(agg.items()):
        print(f"{k:28s} {b:10s} calls {c:4d} total_ms {ms:8.1f} avg_ms {ms / c:7.3f}")
    ins = [(g, d / 1e6) for n_, g, d, _ in rows if "build_insert" in n_]
    print("last 5 descents (grid threads, ms):", ins[-5:])
    if rows:
        print("span_ms", (rows[-1][3] + rows[-1][2] - rows[0][3]) / 1e6)
