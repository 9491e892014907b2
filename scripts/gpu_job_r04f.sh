#!/bin/bash
# round 4, sixth box session: the bench line with in-run counter traffic (two rocprofv3 --pmc child passes), the crossover of the
# quotient-set walk and the bitmap walk between ef_search 450 and 800.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
repo=$PWD
out=$repo/gpurun_out/r04f
mkdir -p $out
cat .build_commit > $out/commit.txt
echo "== bench C3 with in-run traffic"; ( time timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err ) 2>&1 | grep real; python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r04f/bench_c3.json') if l.startswith('{')][-1])
    r=d['roofline']
    print('value',d['value'],'frac',r['frac'],'traffic',r['traffic'],r.get('traffic_over_algorithmic'),json.dumps(r['traffic_in_run'])[:600])
    print('build',d['build']['device_seconds'],d['build']['roofline']['frac'],'parity',d['parity'],'single',d['single_query']['gpu_kernel_ms_median'])
except Exception as e: print('bench parse failed',e); print(open('gpurun_out/r04f/bench_c3.err').read()[-1500:])
PY
echo "== ef crossover"; PB_REPS=2 timeout 900 python scripts/probe_r03_ef.py $out/probe_r04_ef_crossover_c3.jsonl C3 450,550,650 > $out/ef.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04f/probe_r04_ef_crossover_c3.jsonl'):
    d=json.loads(l); print(d['ef'], {k:v for k,v in d.items() if k.endswith('8TBps') or k.endswith('_err')})
PY
du -sh $out
