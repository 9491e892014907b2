#!/bin/bash
# Run-to-run spread of the search kernel at the operating points round 2 could not hold steady: C4 at ef_search 200 in five
# fresh processes, C5 at ef_search 200 in three (each builds its own index; scripts/probe_r03_ef.py, three fresh contexts each).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/${1:-r03l}
mkdir -p $out
for i in 1 2 3 4 5; do timeout 300 python scripts/probe_r03_ef.py $out/spread_c4_ef200.jsonl C4 200 2>&1 | grep '^{' | cut -c1-330; done
for i in 1 2 3; do timeout 600 python scripts/probe_r03_ef.py $out/spread_c5_ef200.jsonl C5 200 2>&1 | grep '^{' | cut -c1-330; done
