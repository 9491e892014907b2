"""Per-step timeline of the pipelined build from a rocprofv3 --kernel-trace db: descent, gap before it, A2, B, B2 (ms), every 8th full step.
usage: trace_timeline.py <dir>"""
import glob
import sqlite3
import sys

f = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]
cur = sqlite3.connect(f).cursor()
rows = cur.execute("select name, grid_x, start, end from kernels where name like '%idist::build%' order by start").fetchall()
full = [(s, e) for n, g, s, e in rows if "build_insert" in n and g // 64 >= 768]
nf = len(full)
sel = [(s, e) for n, g, s, e in rows if "build_select" in n][-nf:]
fast = [(s, e) for n, g, s, e in rows if "build_update_fast" in n][-nf:]
upd = [(s, e) for n, g, s, e in rows if "build_update_kernel" in n][-nf:]
print("span_ms", (rows[-1][3] - rows[0][2]) / 1e6, "full steps", nf, "first full step at ms", (full[0][0] - rows[0][2]) / 1e6)
print("step  A_ms  gap_before  A2_ms  B_ms  B2_ms  A2start-Aend  S_end-A_end(next)")
for i in range(0, nf, 8):
    gap = (full[i][0] - full[i - 1][1]) / 1e6 if i else 0
    nxt = full[i + 1] if i + 1 < nf else full[i]
    print(i, round((full[i][1] - full[i][0]) / 1e6, 2), round(gap, 2), round((sel[i][1] - sel[i][0]) / 1e6, 2),
          round((fast[i][1] - fast[i][0]) / 1e6, 2), round((upd[i][1] - upd[i][0]) / 1e6, 2),
          round((sel[i][0] - full[i][1]) / 1e6, 2), round((upd[i][1] - nxt[1]) / 1e6, 2))
