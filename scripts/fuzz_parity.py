"""Broader seeded fuzz of the parity cases (search in both variants + exact build) on the GPU: the committed tests
run 14 fixed cases; this runs FUZZ_COUNT more from FUZZ_SEED.  Needs the oracle (test infrastructure)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
import instant_distance_amd as ida  # noqa: E402
from test_parity import _fuzz_cases  # noqa: E402

oracle.build_lib()
oracle.lib()
seed, count = int(os.environ.get("FUZZ_SEED", 1000)), int(os.environ.get("FUZZ_COUNT", 40))
bad = 0
for i, c in enumerate(_fuzz_cases("gpu", count, seed)):
    try:
        pc.check_search_parity(ida, oracle, n=c["n"], dim=c["dim"], ef_search=c["ef"], metric=c["metric"], kind=c["kind"],
                               nq=64, seed=c["seed"], ef_construction=c["efc"])      # (strict ties always end with the reference's bytes)
        pc.check_build_exact(ida, oracle, n=min(c["n"], 1200), dim=c["dim"], metric=c["metric"], kind=c["kind"],
                             ef_construction=c["efc"], keep_pruned=c["keep"], seed=c["seed"] + 1)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(json.dumps({"case": i, "cfg": c, "error": repr(e)[:300]}), flush=True)
print(json.dumps({"fuzz_seed": seed, "cases": count, "failed": bad}), flush=True)
sys.exit(1 if bad else 0)
