#!/bin/bash
# round 4, final evidence on the final library: whole GPU suite, smoke, then the headline bench with kernel trace and PMC passes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04_final
mkdir -p $out
cat .build_commit > $out/commit.txt
echo "== pytest -m gpu"; ( time timeout 1150 python -m pytest tests -m gpu -x -q > $out/pytest_gpu_r04_final.log 2>&1 ) 2>&1 | grep real; tail -3 $out/pytest_gpu_r04_final.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/profile_bench.sh r04_final 2>&1 | tail -c 2500
