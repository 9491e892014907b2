#!/bin/bash
# round 3, third box session: the build after step A3 + the reworked MFMA loops (300-d and 768-d), MFMA counters,
# scalar calls from threads with the library's hardware-queue default.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r03c
mkdir -p $out
echo "== build 300"; timeout 600 python scripts/probe_r03_build.py $out/probe_build_300.jsonl default,ids3,r256w4,r256w5,q16w3,a2tile,default2 > $out/probe_build_300.log 2>&1; cut -c1-200 $out/probe_build_300.log
echo "== build 768"; PB_DIM=768 timeout 600 python scripts/probe_r03_build.py $out/probe_build_768.jsonl default,ids3,r256w4,r256w5 > $out/probe_build_768.log 2>&1; cut -c1-200 $out/probe_build_768.log
echo "== quick parity"; timeout 900 python -m pytest tests/test_parity.py -m gpu -x -q -k "build_exact_gpu or bruteforce or batched" > $out/pytest_quick.log 2>&1; tail -2 $out/pytest_quick.log
echo "== threads"; unset GPU_MAX_HW_QUEUES; timeout 300 python scripts/probe_r03_threads.py $out/probe_threads.jsonl default,hwq4 > $out/probe_threads.log 2>&1; head -2 $out/probe_threads.log | cut -c1-300
bash scripts/profile_mfma.sh r03c
