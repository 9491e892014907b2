#!/bin/bash
# round 3, box session g: tie bags and the completion word of narrow host-pointer calls on real hardware.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r03g
mkdir -p $out
echo "== quick parity"; timeout 900 python -m pytest tests/test_parity.py tests/test_shards.py -m gpu -x -q -k "spill or tie_policy or strict_ties or zero_copy or shared_by_threads or grows" > $out/pytest_quick.log 2>&1; tail -3 $out/pytest_quick.log
echo "== threads"; timeout 400 python scripts/probe_r03_threads.py $out/probe_threads.jsonl default,sync_stream > $out/probe_threads.log 2>&1; cat $out/probe_threads.log | cut -c1-300
