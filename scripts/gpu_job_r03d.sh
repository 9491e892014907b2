#!/bin/bash
# round 3, fourth box session: build after the step-B change (300-d / 768-d), counter list, MFMA counters of both MFMA kernels.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r03d
mkdir -p $out
echo "== build 300"; timeout 600 python scripts/probe_r03_build.py $out/probe_build_300.jsonl default,r256w5,q16w3,ids3,default2 > $out/probe_build_300.log 2>&1; cut -c1-150 $out/probe_build_300.log
echo "== build 768"; PB_DIM=768 timeout 600 python scripts/probe_r03_build.py $out/probe_build_768.jsonl default,r256w5,ids3 > $out/probe_build_768.log 2>&1; cut -c1-150 $out/probe_build_768.log
rocprofv3 -L > $out/counters_list.txt 2>&1; grep -i -E "mall|EA0_RDREQ|EA0_WRREQ|TCC_HIT|TCC_MISS|HBM|DRAM" $out/counters_list.txt | cut -c1-160 | head -40
bash scripts/profile_mfma.sh r03d
