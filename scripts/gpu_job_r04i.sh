#!/bin/bash
# round 4: long `nearest` lists merged in one pass by the fat on-chip waves (w_push_merge<16>) — ef_search 650 .. 1000 on C3 by walk,
# and the headline line again (the extra instantiation must not cost the ef 100 kernel anything).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04i
mkdir -p $out
cat .build_commit > $out/commit.txt
echo "== long walk parity"; timeout 600 python -m pytest tests/test_parity.py -m gpu -x -q -k "ef_sweep or merge_width" > $out/pytest.log 2>&1; tail -2 $out/pytest.log
echo "== ef sweep"; PB_REPS=2 timeout 900 python scripts/probe_r03_ef.py $out/probe_r04_ef_paths_wide_merge_c3.jsonl C3 400,650,800,1000 > $out/ef.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04i/probe_r04_ef_paths_wide_merge_c3.jsonl'):
    d=json.loads(l); print(d['ef'], {k:v for k,v in d.items() if k.endswith('8TBps') or k.endswith('_err') or k.endswith('_ms')})
PY
echo "== bench C3"; timeout 600 python bench.py --steps 20 --warmup 5 --no-traffic --threads "" > $out/bench_c3.json 2> $out/bench_c3.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04i/bench_c3.json') if l.startswith('{')][-1]); print('value',d['value'],'frac',d['roofline']['frac'],'kernel_ms',d['roofline']['kernel_ms_avg'],'build',d['build']['device_seconds'],'parity',d['parity']['all_identical'])
PY
