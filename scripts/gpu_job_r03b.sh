#!/bin/bash
# round 3, second box session: scalar calls from T host threads under runtime settings; the build under the descent's
# visited-set form / register budget / waves per CU (300-d and 768-d).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r03b
mkdir -p $out
echo "== threads"; timeout 900 python scripts/probe_r03_threads.py $out/probe_threads.jsonl > $out/probe_threads.log 2>&1; cat $out/probe_threads.log | cut -c1-300
echo "== build 300"; timeout 600 python scripts/probe_r03_build.py $out/probe_build_300.jsonl > $out/probe_build_300.log 2>&1; cut -c1-260 $out/probe_build_300.log
echo "== build 768"; PB_DIM=768 timeout 600 python scripts/probe_r03_build.py $out/probe_build_768.jsonl default,ids3,r256w4,r256w5,r256w6,onestream > $out/probe_build_768.log 2>&1; cut -c1-260 $out/probe_build_768.log
