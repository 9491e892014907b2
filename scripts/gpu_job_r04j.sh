#!/bin/bash
# round 4: the speculative overflow test-and-set once more, now on top of the 1024-entry one-pass merge (same probe as r04i).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04j
mkdir -p $out
cat .build_commit > $out/commit.txt
PB_REPS=2 timeout 900 python scripts/probe_r03_ef.py $out/probe_r04_ef_paths_wide_merge_plus_speculation_c3.jsonl C3 200,400,650,800,1000 > $out/ef.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04j/probe_r04_ef_paths_wide_merge_plus_speculation_c3.jsonl'):
    d=json.loads(l); print(d['ef'], {k:v for k,v in d.items() if k.endswith('8TBps') or k.endswith('_err') or k=='q16_ms' or k=='q16_same_ids'})
PY
