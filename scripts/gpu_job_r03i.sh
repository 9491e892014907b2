#!/bin/bash
# round 3, box session i: scalar calls from many threads through the launch combiner.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r03i
mkdir -p $out
echo "== threads parity"; timeout 600 python -m pytest tests/test_shards.py tests/test_parity.py -m gpu -x -q -k "scalar_calls or shared_by_threads or zero_copy" > $out/pytest_quick.log 2>&1; tail -3 $out/pytest_quick.log
echo "== threads"; timeout 400 python scripts/probe_r03_threads.py $out/probe_threads.jsonl default,no_combine > $out/probe_threads.log 2>&1; cat $out/probe_threads.log | cut -c1-500
