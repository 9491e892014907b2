#!/bin/bash
# round 4: long walks on two thinner waves per SIMD (q16w2) against one fat wave per SIMD (q16), the bitmap walk and the policy.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04k
mkdir -p $out
cat .build_commit > $out/commit.txt
PB_REPS=2 timeout 900 python scripts/probe_r03_ef.py $out/probe_r04_ef_paths_two_waves_per_simd_c3.jsonl C3 200,300,400,650,800,1000 > $out/ef.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04k/probe_r04_ef_paths_two_waves_per_simd_c3.jsonl'):
    d=json.loads(l); print(d['ef'], {k:v for k,v in d.items() if k.endswith('8TBps') or k.endswith('_err') or k.endswith('same_ids') and not v})
PY
tail -3 $out/ef.log | cut -c1-300
