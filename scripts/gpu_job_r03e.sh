#!/bin/bash
# round 3, fifth box session: the judged evidence — C3 bench + rocprofv3 trace + PMC passes (profile_bench.sh), then the C4 and C5
# bench lines with --check.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r03e
mkdir -p $out
bash scripts/profile_bench.sh r03e 2>&1 | tail -40
echo "== bench C4"; timeout 900 python bench.py --config C4 --check --steps 5 --warmup 2 > $out/bench_c4.json 2> $out/bench_c4.err; echo rc=$?; head -c 300 $out/bench_c4.json
echo "== bench C5"; timeout 1500 python bench.py --config C5 --check --steps 5 --warmup 2 > $out/bench_c5.json 2> $out/bench_c5.err; echo rc=$?; head -c 300 $out/bench_c5.json
echo "== bench C2"; timeout 600 python bench.py --config C2 --check --steps 5 --warmup 2 > $out/bench_c2.json 2> $out/bench_c2.err; echo rc=$?; head -c 300 $out/bench_c2.json
du -sh gpurun_out
