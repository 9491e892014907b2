#!/bin/bash
# round 4, fifth box session: Gram-matrix selection with the verdict masks in registers (exact-build tests, MFMA counters), C3
# build, runtime-geometry builds with their new default (512-register descents).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
repo=$PWD
out=$repo/gpurun_out/r04e
mkdir -p $out
cat .build_commit > $out/commit.txt
echo "== exact build tests"; timeout 900 python -m pytest tests/test_parity.py -m gpu -x -q -k "build_exact_gpu or lattice or fuzz_search_and_exact or batched" > $out/pytest_build.log 2>&1; tail -3 $out/pytest_build.log
echo "== C3 build"; PB_REPS=3 timeout 600 python scripts/probe_r04_build.py $out/probe_r04_build_c3.jsonl default > $out/build_c3.log 2>&1; cut -c1-330 $out/probe_r04_build_c3.jsonl
for d in 1024 384; do
  echo "== build dim $d"; PB_DIM=$d PB_REPS=1 timeout 900 python scripts/probe_r04_build.py $out/probe_r04_build_dim$d.jsonl default > $out/build_dim$d.log 2>&1; cut -c1-330 $out/probe_r04_build_dim$d.jsonl
done
bash scripts/profile_mfma.sh r04e build 2>&1 | tail -25 | cut -c1-400
du -sh $out
