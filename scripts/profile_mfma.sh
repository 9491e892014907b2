#!/bin/bash
# MFMA evidence (round 3): the two kernels that put the matrix cores on a path of this engine —
#   build_select_mfma_kernel  select_heuristic of every new point (Gram matrix filter), inside the C3 build
#   mfma_dist_kernel          the exhaustive -2QP^T filter (C4's distance path / recall ground truth)
# kernel trace + two PMC passes each (counters in their own runs, no other trace domains).
# Usage: scripts/profile_mfma.sh <tag> [build|bf]   -> gpurun_out/<tag>/mfma/
set -u
tag=${1:-r03}
only=${2:-}
repo="${GRAFT_REPO_ROOT:-/root/repo}"
out=$repo/gpurun_out/$tag/mfma
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for case in build:scripts/build_only.py bf:scripts/mfma_case.py; do
  nm=${case%%:*}; py=$repo/${case##*:}
  if [ -n "$only" ] && [ "$only" != "$nm" ]; then continue; fi
  echo "== $nm trace"; timeout 600 rocprofv3 --kernel-trace --stats -d $out/${nm}_trace -o t -- python $py > $out/${nm}_trace.log 2>&1; echo rc=$?; tail -1 $out/${nm}_trace.log | cut -c1-200
  echo "== $nm pmc1"; timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $out/${nm}_pmc1 -o p -- python $py > $out/${nm}_pmc1.log 2>&1; echo rc=$?
  echo "== $nm pmc2"; timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $out/${nm}_pmc2 -o p -- python $py > $out/${nm}_pmc2.log 2>&1; echo rc=$?
done
python $repo/scripts/summarize_mfma.py $out > $out/summary.json 2> $out/summary.err; head -c 3000 $out/summary.json
find $out \( -name "*.db" -o -name "*.csv" \) -size +1M -delete   # only the summary travels back (gpurun_out is capped at 64 MiB)
du -sh $out
