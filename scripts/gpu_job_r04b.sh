#!/bin/bash
# round 4, second box session: early-abandon probe on the headline kernel (times + FETCH_SIZE per variant), the build's stream
# layouts / step caps A-B, a kernel trace of the build (who is the critical resource), GPU tests of the build schedule.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
repo=$PWD
out=$repo/gpurun_out/r04b
mkdir -p $out
cat .build_commit > $out/commit.txt
echo "== build schedule tests"; timeout 900 python -m pytest tests/test_parity.py -m gpu -x -q -k "batched or schedule or c3_full or build_exact_gpu" > $out/pytest_build.log 2>&1; tail -3 $out/pytest_build.log
echo "== early abandon probe"; timeout 600 python scripts/probe_r04_ea.py $out/probe_r04_early_abandon.jsonl > $out/ea.log 2>&1; cat $out/probe_r04_early_abandon.jsonl | cut -c1-400
echo "== build schedule probe"; PB_REPS=3 timeout 900 python scripts/probe_r04_build.py $out/probe_r04_build_schedule.jsonl default,r03_schedule,two_descent_streams_only,selection_stream_only,cap16384,cap32768,check,default > $out/build_probe.log 2>&1; python - <<'PY'
import json
for l in open('gpurun_out/r04b/probe_r04_build_schedule.jsonl'):
    d=json.loads(l); print(d.get('case'), d.get('seconds'), d.get('frac_of_8TBps'), d.get('recall_at_10'), d.get('graph_checksum'), d.get('err'))
PY
cd /tmp && export TMPDIR=/tmp
echo "== early abandon FETCH_SIZE"; timeout 600 rocprofv3 --pmc FETCH_SIZE -d $out/ea_pmc -o ea -- python $repo/scripts/probe_r04_ea.py > $out/ea_pmc.log 2>&1; echo "rc=$?"
python - $out <<'PY'
import glob, sqlite3, sys, json, collections
out = sys.argv[1]
try:
    f = glob.glob(out + '/ea_pmc/**/*_results.db', recursive=True)[0]
    cur = sqlite3.connect(f).cursor()
    agg = collections.defaultdict(list)
    for n, v, d in cur.execute("select kernel_name, value, duration from counters_collection where counter_name='FETCH_SIZE' and kernel_name like '%search_kernel%'"):
        agg[n.split('search_kernel')[1].split('>')[0]].append((v, d))
    res = {}
    for k, v in agg.items():
        big = [x for x in v if x[0] > 0.5 * max(y[0] for y in v)]        # the 10k-query launches
        res[k] = {"launches": len(big), "FETCH_SIZE_KB_mean": round(sum(x[0] for x in big) / len(big), 1), "ms_mean": round(sum(x[1] for x in big) / len(big) / 1e6, 3)}
    print(json.dumps(res))
    json.dump(res, open(out + '/probe_r04_early_abandon_fetch_size.json', 'w'), indent=1)
except Exception as e:
    print('pmc parse failed', repr(e))
PY
echo "== build kernel trace"; timeout 600 rocprofv3 --kernel-trace --stats -d $out/build_trace -o build -- python $repo/scripts/build_only.py > $out/build_trace.log 2>&1; echo "rc=$?"; tail -1 $out/build_trace.log
python $repo/scripts/trace_busy.py $out/build_trace | tee $out/build_busy.json
python $repo/scripts/trace_stats.py $out/build_trace | head -8
IDIST_BUILD_A_STREAMS=1 IDIST_BUILD_A2_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $out/build_trace_r03 -o build -- python $repo/scripts/build_only.py > $out/build_trace_r03.log 2>&1; echo "rc=$?"
python $repo/scripts/trace_busy.py $out/build_trace_r03 | tee $out/build_busy_r03_schedule.json
# keep the PMC csv of the EA probe if small; drop the big databases
find $out \( -name "*.db" \) -size +1M -delete
du -sh $out
