#!/bin/bash
# round 4: the whole -m gpu suite and the judged C3 profile on the library with the long-walk changes (one-pass merge to 1024 entries,
# two thinner waves per SIMD from ef_search 512 on), plus the spread of the ef 800 operating point over fresh processes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
repo=$PWD
out=$repo/gpurun_out/r04m
mkdir -p $out
cat .build_commit > $out/commit.txt
echo "== pytest gpu"; ( time timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1 ) 2>&1 | grep real; tail -3 $out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/profile_bench.sh r04m 2>&1 | grep -E "^==|rc=" | head -12
python - <<'PY'
import json
s=json.load(open('gpurun_out/r04m/summary.json')); print('agreement', s.get('agreement'), s.get('agreement_error'))
j=s['bench']; r=j['roofline']; print('bench', j['commit'], j['value'], r['frac'], r.get('traffic_over_algorithmic'), 'build', j['build']['device_seconds'], j['build']['roofline']['frac'], j['single_query']['gpu_kernel_ms_median'])
PY
echo "== ef 800 over three fresh processes"; for i in 1 2 3; do PB_REPS=1 timeout 300 python scripts/probe_r03_ef.py $out/probe_r04_ef800_fresh_processes_c3.jsonl C3 800 > $out/ef800_$i.log 2>&1; done
python - <<'PY'
import json
for l in open('gpurun_out/r04m/probe_r04_ef800_fresh_processes_c3.jsonl'):
    d=json.loads(l); print({k:v for k,v in d.items() if k.endswith('_ms') or k.endswith('8TBps')})
PY
du -sh $out
