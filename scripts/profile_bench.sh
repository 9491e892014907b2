#!/bin/bash
# One GPU-box session: bench line, rocprofv3 kernel-trace stats of the same command, and separate
# PMC passes (FETCH_SIZE / WRITE_SIZE) for the bench and for the calibration kernel.
# Usage: scripts/profile_bench.sh <tag>   (writes under gpurun_out/<tag>/)
set -u
tag=${1:-r01}
repo="${GRAFT_REPO_ROOT:-/root/repo}"
out=$repo/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
echo "== bench"; timeout 600 python $repo/bench.py > $out/bench.json 2> $out/bench.err; echo "rc=$?"; tail -c 3000 $out/bench.json
echo "== kernel trace"; timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python $repo/bench.py --no-cpu-baseline --no-traffic --threads "" > $out/trace.log 2>&1; echo "rc=$?"
echo "== pmc FETCH_SIZE"; timeout 900 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o bench -- python $repo/bench.py --no-cpu-baseline --no-traffic --steps 3 --threads "" > $out/pmc_fetch.log 2>&1; echo "rc=$?"
echo "== pmc WRITE_SIZE"; timeout 900 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o bench -- python $repo/bench.py --no-cpu-baseline --no-traffic --steps 3 --threads "" > $out/pmc_write.log 2>&1; echo "rc=$?"
echo "== pmc TCC (L2 hit rate, fabric reads and how many of them went to the DRAM controllers)"; timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum -d $out/pmc_tcc -o bench -- python $repo/bench.py --no-cpu-baseline --no-traffic --steps 3 --threads "" > $out/pmc_tcc.log 2>&1; echo "rc=$?"
echo "== calib FETCH_SIZE"; timeout 600 rocprofv3 --pmc FETCH_SIZE -d $out/calib_fetch -o calib -- python $repo/scripts/calib_fetch.py > $out/calib_fetch.log 2>&1; echo "rc=$?"; tail -2 $out/calib_fetch.log
python $repo/scripts/summarize_profile.py $out $out/rocprof_summary_$tag > $out/summary.json 2> $out/summary.err; cat $out/summary.json | head -c 4000
# only summaries travel back (gpurun_out is capped at 64 MiB): drop the raw databases / traces
find $out \( -name "*.db" -o -name "*.csv" \) -size +1M -delete
du -sh $out
