#!/bin/bash
# round 4, third box session: the stream layout in narrow steps (C2-size and C3 builds), where a 1024-d build spends its time,
# and the parity tests that now route the test-only walk variants through libidist_variants.so.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
repo=$PWD
out=$repo/gpurun_out/r04c
mkdir -p $out
cat .build_commit > $out/commit.txt
echo "== variant tests"; timeout 900 python -m pytest tests/test_parity.py -m gpu -x -q -k "search_parity or schedule or heavy_ties or build_exact_gpu or spill" > $out/pytest_variants.log 2>&1; tail -3 $out/pytest_variants.log
echo "== C2-size build"; PB_N=100000 PB_DIM=128 PB_REPS=5 timeout 600 python scripts/probe_r04_build.py $out/probe_r04_build_schedule_c2.jsonl default,streams_off,streams_all,no_quad,default > $out/build_probe_c2.log 2>&1
echo "== C3 build"; PB_REPS=3 timeout 600 python scripts/probe_r04_build.py $out/probe_r04_build_schedule_c3.jsonl default,streams_off,default > $out/build_probe_c3.log 2>&1
python - <<'PY'
import json
for f in ('c2','c3'):
    for l in open(f'gpurun_out/r04c/probe_r04_build_schedule_{f}.jsonl'):
        d=json.loads(l); print(f, d.get('case'), d.get('seconds'), d.get('frac_of_8TBps'), d.get('recall_at_10'), d.get('graph_checksum'), d.get('err'))
PY
cd /tmp && export TMPDIR=/tmp
for d in 1024 384; do
  echo "== build kernel trace dim $d"; PB_DIM=$d timeout 600 rocprofv3 --kernel-trace --stats -d $out/build_trace_$d -o build -- python $repo/scripts/build_only.py > $out/build_trace_$d.log 2>&1; echo "rc=$?"; tail -1 $out/build_trace_$d.log
  python $repo/scripts/trace_busy.py $out/build_trace_$d | tee $out/build_busy_dim$d.json
  python $repo/scripts/trace_stats.py $out/build_trace_$d | head -7
  python $repo/scripts/trace_steps.py $out/build_trace_$d | head -20
done
find $out \( -name "*.db" \) -size +1M -delete
du -sh $out
