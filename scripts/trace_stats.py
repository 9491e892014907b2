"""Print the top kernels of a rocprofv3 --kernel-trace --stats output directory.  usage: trace_stats.py <dir>"""
import glob
import os
import sqlite3
import sys

for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)):
    cur = sqlite3.connect(f).cursor()
    for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()[:12]:
        print(f"{r[0][:80]:80s} calls {r[1]:5d} total_ms {r[2] / 1e6:9.2f} avg_us {r[3] / 1e3:9.1f} {r[4]:5.1f}%")
