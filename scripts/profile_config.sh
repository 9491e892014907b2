#!/bin/bash
# rocprofv3 --kernel-trace --stats of one BASELINE configuration's bench line (C4 / C5): top kernels, the search kernel's
# full-batch launches against the bench's own HIP events.  Usage: scripts/profile_config.sh <tag> <C4|C5>
set -u
tag=$1; cfg=$2
repo="${GRAFT_REPO_ROOT:-/root/repo}"
out=$repo/gpurun_out/$tag/$cfg
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python $repo/bench.py --config $cfg --no-cpu-baseline --no-traffic --threads "" --steps 5 --warmup 2 > $out/trace.log 2>&1; echo "rc=$?"
python $repo/scripts/summarize_profile.py $out $out/rocprof_summary_${tag}_$cfg "\`python bench.py --config $cfg --no-cpu-baseline --steps 5 --warmup 2\`" > $out/summary.json 2> $out/summary.err
python - <<PY
import json
s = json.load(open("$out/summary.json"))
print(json.dumps({"agreement": s.get("agreement"), "top": s.get("kernel_stats", [])[:4]})[:1500])
PY
find $out \( -name "*.db" -o -name "*.csv" \) -size +1M -delete
