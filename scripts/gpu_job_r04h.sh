#!/bin/bash
# round 4, evidence session on the final library: the judged C3 profile (bench line with in-run traffic + rocprofv3 kernel trace +
# separate PMC passes + calibration), the other BASELINE configurations and two non-template dimensions through bench.py, MFMA counters.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
repo=$PWD
out=$repo/gpurun_out/r04h
mkdir -p $out
cat .build_commit > $out/commit.txt
bash scripts/profile_bench.sh r04h 2>&1 | grep -E "^==|rc=|agreement" | head -20
for c in C2 C4 C5; do
  echo "== bench $c"; ( time timeout 1500 python bench.py --config $c --check > $out/bench_$c.json 2> $out/bench_$c.err ) 2>&1 | grep real
done
for d in 384 1024; do
  echo "== bench dim $d"; timeout 900 python bench.py --dim $d --steps 10 --warmup 2 --threads "" --cpu-build-sample 0 > $out/bench_dim$d.json 2> $out/bench_dim$d.err; echo "rc=$?"
done
python - <<'PY'
import json
for f in ('bench','bench_C2','bench_C4','bench_C5','bench_dim384','bench_dim1024'):
    try:
        j=json.loads([l for l in open(f'gpurun_out/r04h/{f}.json') if l.startswith('{')][-1])
        r=j['roofline']
        print(f, j['commit'], 'value',j['value'],'frac',r['frac'],'ef',j['config']['ef_search'],'recall',j['config']['recall_at_10'],'traffic/alg',r.get('traffic_over_algorithmic'),
              'build',j['build']['device_seconds'],j['build']['roofline']['frac'],'parity',(j.get('parity') or {}).get('all_identical'),'checks',all(v for k,v in (j.get('checks') or {}).items() if isinstance(v,bool)),
              'single',j['single_query'].get('gpu_kernel_ms_median'), 'thr16', (j['single_query'].get('threads') or {}).get('16',{}).get('gpu_calls_per_s'))
    except Exception as e: print(f,'parse failed',repr(e))
PY
bash scripts/profile_mfma.sh r04h 2>&1 | grep -E "^==|rc=" | head
du -sh $out
