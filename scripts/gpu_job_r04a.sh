#!/bin/bash
# round 4, first box session: the new parity tests + the tests the round's kernel changes touch, the C3 bench line, the step-B
# look-up probe (measurement build), build A/B of the descent-stream schedule, and bench lines at two non-template dimensions.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04a
mkdir -p $out
cat .build_commit > $out/commit.txt
echo "== pytest subset"; timeout 1500 python -m pytest tests/test_scale_gpu.py tests/test_parity.py -m gpu -x -q -k "scale or large or runtime_geometry or build_exact_gpu or fuzz or heavy_ties or tie or batched or schedule or c3_full or duplicate or ef_sweep or search_parity" > $out/pytest_subset.log 2>&1; tail -4 $out/pytest_subset.log
echo "== bench C3"; timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err; echo "rc=$?"; python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r04a/bench_c3.json') if l.startswith('{')][-1])
    print('value',d['value'],'frac',d['roofline']['frac'],'build',d['build']['device_seconds'],d['build']['roofline']['frac'],'parity',d['parity'])
except Exception as e: print('bench parse failed',e)
PY
echo "== probe step B"; timeout 300 python scripts/probe_build_lib.py libidist_probe.so > $out/probe_stepB.log 2>&1; tail -4 $out/probe_stepB.log | cut -c1-700
echo "== build A/B: one descent stream"; timeout 300 python scripts/probe_build_lib.py libidist.so IDIST_BUILD_A_STREAMS=1 > $out/build_one_stream.log 2>&1; tail -2 $out/build_one_stream.log
echo "== build: two descent streams"; timeout 300 python scripts/probe_build_lib.py libidist.so > $out/build_two_streams.log 2>&1; tail -2 $out/build_two_streams.log
for d in 384 1024; do
  echo "== bench dim $d"; timeout 900 python bench.py --dim $d --steps 10 --warmup 2 --threads "" --cpu-build-sample 0 > $out/bench_dim$d.json 2> $out/bench_dim$d.err; echo "rc=$?"
  python - $d <<'PY'
import json,sys
d=sys.argv[1]
try:
    j=json.loads([l for l in open(f'gpurun_out/r04a/bench_dim{d}.json') if l.startswith('{')][-1])
    print('dim',d,'value',j['value'],'frac',j['roofline']['frac'],'ef',j['config']['ef_search'],'recall',j['config']['recall_at_10'],'build',j['build']['device_seconds'],j['build']['roofline']['frac'],'parity',j['parity'])
except Exception as e: print('parse failed',e)
PY
done
du -sh $out
