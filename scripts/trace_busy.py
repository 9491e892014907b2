"""Who is the critical resource of the pipelined build?  From a rocprofv3 --kernel-trace db: for the LAST build in the trace,
the union of the intervals each kernel class is running (ms), pairwise overlaps, and the time nothing of a class runs.
usage: trace_busy.py <dir>"""
import glob
import json
import sqlite3
import sys

f = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]
cur = sqlite3.connect(f).cursor()
rows = cur.execute("select name, start, end from kernels where name like '%idist::%' order by start").fetchall()
# the last build = from the last permute_rows_kernel launch on
starts = [s for n, s, e in rows if "permute_rows" in n]
t0 = starts[-1] if starts else rows[0][1]
rows = [(n, s, e) for n, s, e in rows if s >= t0 and ("idist::build" in n or "copy_rows" in n or "snapshot" in n)]
cls = {"descents": "build_insert", "selection": "build_select", "updates": "build_update_fast", "full_updates": "build_update_kernel", "carry": "copy_rows"}


def union(iv):
    iv = sorted(iv)
    out, cs, ce = [], None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            out.append((cs, ce)); cs, ce = s, e
    if cs is not None:
        out.append((cs, ce))
    return out


def length(u):
    return sum(e - s for s, e in u) / 1e6


def inter(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot / 1e6


span = (max(e for _, _, e in rows) - min(s for _, s, _ in rows)) / 1e6
U = {k: union([(s, e) for n, s, e in rows if v in n]) for k, v in cls.items()}
S = union([(s, e) for n, s, e in rows if "build_insert" not in n])
res = {"span_ms": round(span, 2), "launches": len(rows),
       "busy_ms": {k: round(length(u), 2) for k, u in U.items()}, "sum_ms": {k: round(sum(e - s for n, s, e in rows if v in n) / 1e6, 2) for k, v in cls.items()},
       "update_side_busy_ms": round(length(S), 2), "descents_idle_ms": round(span - length(U["descents"]), 2), "update_side_idle_ms": round(span - length(S), 2),
       "descents_and_update_side_overlap_ms": round(inter(U["descents"], S), 2), "selection_and_updates_overlap_ms": round(inter(U["selection"], U["updates"]), 2),
       "two_descent_launches_overlap_ms": round(sum(e - s for n, s, e in rows if "build_insert" in n) / 1e6 - length(U["descents"]), 2)}
print(json.dumps(res))
