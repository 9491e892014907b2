#!/bin/bash
# round 3, final box session: the whole -m gpu suite (C4 / C5 full-size tests included), then the judged C3 profile again with
# the final library, and rocprofv3 kernel-trace summaries of the C4 and C5 bench lines.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r03k
mkdir -p $out
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -5 $out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/profile_bench.sh r03k 2>&1 | grep -E "^==|rc=|agreement" | head -20
bash scripts/profile_config.sh r03k C4 2>&1 | tail -3 | cut -c1-600
bash scripts/profile_config.sh r03k C5 2>&1 | tail -3 | cut -c1-600
du -sh gpurun_out
