#!/bin/bash
# round 3, first box session: environment, quick parity of the new paths, ef / register-budget probes, the three bench
# lines (C3, C4, C5 with --check), then the whole -m gpu suite.  Output under gpurun_out/r03a/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r03a
mkdir -p $out
{ free -g; nproc; cat /sys/fs/cgroup/cpu.max; rocm-smi --showmeminfo vram 2>/dev/null | head -8; } > $out/env.txt 2>&1
echo "== quick"; timeout 900 python -m pytest tests/test_shards.py tests/test_parity.py -m gpu -x -q -k "replicate or search_parity or grows or ties" > $out/pytest_quick.log 2>&1; tail -3 $out/pytest_quick.log
echo "== probe ef C3"; timeout 600 python scripts/probe_r03_ef.py $out/probe_ef_c3.jsonl C3 > $out/probe_ef_c3.log 2>&1; tail -2 $out/probe_ef_c3.log | cut -c1-400
echo "== probe tune"; timeout 400 python scripts/probe_r03_tune.py $out/probe_tune.jsonl 100,200 > $out/probe_tune.log 2>&1; tail -3 $out/probe_tune.log
echo "== bench C3"; timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_c3.json 2> $out/bench_c3.err; echo rc=$?; head -c 600 $out/bench_c3.json
echo "== bench C4"; timeout 900 python bench.py --config C4 --check --steps 5 --warmup 2 > $out/bench_c4.json 2> $out/bench_c4.err; echo rc=$?; head -c 600 $out/bench_c4.json
echo "== bench C5"; timeout 1500 python bench.py --config C5 --check --steps 5 --warmup 2 > $out/bench_c5.json 2> $out/bench_c5.err; echo rc=$?; head -c 600 $out/bench_c5.json; tail -5 $out/bench_c5.err
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_configs_gpu.py > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log
