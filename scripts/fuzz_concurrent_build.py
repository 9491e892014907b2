"""Seeded fuzz of the default (concurrent, pipelined) build on the GPU: row invariants, determinism, recall vs the
oracle's threaded build.  Needs the oracle (test infrastructure)."""
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
seed, count = int(os.environ.get("FUZZ_SEED", 3000)), int(os.environ.get("FUZZ_COUNT", 16))
bad = 0
for i, c in enumerate(_fuzz_cases("gpu", count, seed)):
    n = c["n"] * (12 if c["dim"] <= 128 else 6)
    try:
        rec, orec = pc.check_build_concurrent_invariants(ida, oracle, n=n, dim=c["dim"], kind=c["kind"], metric=c["metric"],
                                                         ef_construction=max(c["efc"], 40), keep_pruned=c["keep"], seed=c["seed"])
        print(json.dumps({"case": i, "n": n, "dim": c["dim"], "kind": c["kind"], "efc": max(c["efc"], 40), "recall": round(rec, 4),
                          "oracle_threaded_recall": round(orec, 4)}), flush=True)
    except ida.IdistError as e:
        if e.status == 6 and c["kind"] == "grid" and c["dim"] <= 8:    # documented limit: reported, never silent
            print(json.dumps({"case": i, "n": n, "dim": c["dim"], "kind": c["kind"], "tie_overflow_reported": True}), flush=True)
            continue
        bad += 1
        print(json.dumps({"case": i, "cfg": c, "n": n, "error": repr(e)[:300]}), flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(json.dumps({"case": i, "cfg": c, "n": n, "error": repr(e)[:300]}), flush=True)
print(json.dumps({"fuzz_seed": seed, "cases": count, "failed": bad}), flush=True)
sys.exit(1 if bad else 0)
