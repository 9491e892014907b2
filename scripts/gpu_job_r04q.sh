#!/bin/bash
# round 4: four-wave walk with the exact next candidate — short parity pass, segment probe, one query per call / threads, C2-size build
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/${1:-r04q}
mkdir -p $out
cat .build_commit > $out/commit.txt
echo "== parity"; ( time timeout 300 python -m pytest tests/test_parity.py tests/test_host_cpp.py -m gpu -x -q -k "scalar or threads or build_exact_small or c1_ or host" > $out/pytest_quad.log 2>&1 ) 2>&1 | grep real; tail -3 $out/pytest_quad.log
echo "== probe"; timeout 300 python scripts/probe_r04_quad.py $out/probe_r04_quad_segments.jsonl > $out/probe.log 2>&1; cat $out/probe_r04_quad_segments.jsonl
echo "== bench C3"; timeout 600 python bench.py --steps 10 --warmup 3 --no-traffic > $out/bench_c3.json 2> $out/bench_c3.err; OUT=$out python - <<'PY'
import json, os
d=json.loads([l for l in open(os.environ['OUT']+'/bench_c3.json') if l.startswith('{')][-1]); print('value',d['value'],'frac',d['roofline']['frac'],'build',d['build']['device_seconds'],'parity',d['parity']['all_identical'])
print('single', json.dumps(d['single_query'])[:900])
PY
echo "== C2-size build"; PB_N=100000 PB_DIM=128 PB_REPS=3 timeout 300 python scripts/probe_r04_build.py $out/probe_r04_build_c2.jsonl default > $out/build_c2.log 2>&1; cut -c1-260 $out/probe_r04_build_c2.jsonl
du -sh $out
