"""Sum one counter per kernel over a rocprofv3 --pmc output directory (rocpd sqlite): {kernel: {"dispatches": n, "<counter>": sum}}.
usage: python scripts/pmc_sum.py <dir> [out.json]"""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

out = defaultdict(lambda: defaultdict(float))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)):
    cur = sqlite3.connect(f).cursor()
    for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection").fetchall():
        name = k.split("(")[0].replace("void ", "").replace("idist::", "")
        name = name.split("<")[0] + ("<" + name.split("<", 1)[1] if name.startswith("calib_") and "<" in name else "")
        out[name][c] += v
        out[name]["dispatches:" + c] += 1
res = {k: {kk: round(vv) for kk, vv in v.items()} for k, v in sorted(out.items())}
print(json.dumps(res, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
