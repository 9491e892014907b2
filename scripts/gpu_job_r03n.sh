#!/bin/bash
# round 3, closing box session: the whole -m gpu suite and the default bench line on the final tree.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r03n
mkdir -p $out
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $out/pytest_gpu.log | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err; echo rc=$?; head -c 400 $out/bench_c3.json
echo "== native threads"; g++ -std=c++17 -O2 tests/host/threads.cpp -o /tmp/host_threads -Linstant-distance_amd/csrc -lidist -Wl,-rpath,$PWD/instant-distance_amd/csrc -pthread && for T in 1 8 16 64; do /tmp/host_threads 1000000 300 $T 200 | head -1; done > $out/native_threads.txt 2>&1; cat $out/native_threads.txt
