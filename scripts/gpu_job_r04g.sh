#!/bin/bash
# round 4, validation session: the whole -m gpu suite on the tree of this commit, smoke(), the in-process RCCL replication child of
# bench.py on one device.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r04g
mkdir -p $out
cat .build_commit > $out/commit.txt
echo "== pytest gpu"; ( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > $out/pytest_gpu.log 2>&1 ) 2>&1 | grep real; tail -22 $out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== rccl child"; timeout 300 python bench.py --rccl-child 1 --n 200000 2>&1 | tail -2 | cut -c1-600
du -sh $out
