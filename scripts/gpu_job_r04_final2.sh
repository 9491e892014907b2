#!/bin/bash
# round 4: the headline evidence again after the growth-phase rule changed (host schedule only; kernels as in r04_final)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04_final2
mkdir -p $out
cat .build_commit > $out/commit.txt
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
bash scripts/profile_bench.sh r04_final2 2>&1 | grep -E "^==|rc=|agreement" | head -20
echo "== C2"; timeout 200 python bench.py --config C2 --check --no-traffic --threads 16 > $out/bench_c2.json 2> $out/bench_c2.err; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_final2/bench_c2.json') if l.startswith('{')][-1]); print('C2', d['value'], d['build']['device_seconds'], d['parity']['all_identical'], d['config'].get('recall_at_10'))
d=json.loads([l for l in open('gpurun_out/r04_final2/bench.json') if l.startswith('{')][-1]); print('C3', d['commit'], d['value'], d['roofline']['frac'], d['roofline']['traffic_over_algorithmic'], d['build']['device_seconds'], d['build']['roofline']['frac'], d['parity']['all_identical'], d['config'].get('recall_at_10'))
PY
