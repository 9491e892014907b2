"""Print per-dispatch counter values of one kernel from a rocprofv3 --pmc output directory (rocpd sqlite).
usage: python scripts/pmc_dump.py <dir> [kernel-substring]"""
import glob
import json
import os
import sqlite3
import sys
from collections import OrderedDict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "search_kernel"
for f in sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)):
    cur = sqlite3.connect(f).cursor()
    rows = cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection order by dispatch_id").fetchall()
    disp = OrderedDict()
    for did, k, c, v in rows:
        if sub in k:
            disp.setdefault(did, {})[c] = v
    for i, (did, cs) in enumerate(disp.items()):
        print(json.dumps({"i": i, **{k: round(v) for k, v in cs.items()}}))
