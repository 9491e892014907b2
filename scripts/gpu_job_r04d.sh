#!/bin/bash
# round 4, fourth box session: long walks with the speculative overflow test-and-set (C3, ef_search 200 .. 800: quotient set vs
# bitmap walk vs what the policy picks), parity of those walks at scale, runtime-geometry builds (tile sized to the LDS beside
# the descents; descent register budgets).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
repo=$PWD
out=$repo/gpurun_out/r04d
mkdir -p $out
cat .build_commit > $out/commit.txt
echo "== parity of long walks"; timeout 1200 python -m pytest tests/test_parity.py tests/test_scale_gpu.py -m gpu -x -q -k "ef_sweep or merge_width or c3_full or runtime_geometry or heavy_ties or spill or fuzz_search" > $out/pytest_long_walks.log 2>&1; tail -3 $out/pytest_long_walks.log
echo "== ef sweep C3"; PB_REPS=3 timeout 900 python scripts/probe_r03_ef.py $out/probe_r04_ef_paths_c3.jsonl C3 200,400,800 > $out/ef.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04d/probe_r04_ef_paths_c3.jsonl'):
    d=json.loads(l); print(d['ef'], {k:v for k,v in d.items() if k.endswith('_ms') or k.endswith('8TBps') or k.endswith('same_ids') or k.endswith('_err')})
PY
for d in 1024 384; do
  echo "== build dim $d"; PB_DIM=$d PB_REPS=1 timeout 900 python scripts/probe_r04_build.py $out/probe_r04_build_dim$d.jsonl default,regs512,regs512_w3 > $out/build_dim$d.log 2>&1
  python - $d <<'PY'
import json,sys
for l in open(f'gpurun_out/r04d/probe_r04_build_dim{sys.argv[1]}.jsonl'):
    d=json.loads(l); print(sys.argv[1], d.get('case'), d.get('seconds'), d.get('frac_of_8TBps'), d.get('recall_at_10'), d.get('graph_checksum'), d.get('err'))
PY
done
du -sh $out
