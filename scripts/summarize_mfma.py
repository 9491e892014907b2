"""Condense scripts/profile_mfma.sh's rocprofv3 outputs: per kernel calls / total and average duration (ms) from the kernel
trace, PMC sums, and the derived MFMA figures.  usage: summarize_mfma.py gpurun_out/<tag>/mfma"""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

out = sys.argv[1]
KERNELS = ("build_select_mfma_kernel", "mfma_dist_kernel", "build_insert_kernel", "build_update_fast_kernel",
           "build_update_kernel", "copy_rows_kernel", "rerank_kernel")
CLOCK_GHZ = 2.4          # MI355X_MICROARCH.md: max clock; effective clock = GRBM_GUI_ACTIVE / kernel time where collected
N_SIMD = 256 * 4


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def dbs(sub):
    return sorted(glob.glob(os.path.join(out, sub, "**", "*_results.db"), recursive=True))


res = {}
for case in ("build", "bf"):
    r = {}
    for f in dbs(case + "_trace"):
        cur = sqlite3.connect(f).cursor()
        rows = cur.execute("select name,total_calls,total_duration,average from top_kernels").fetchall()
        # top_kernels durations are MICROseconds (the 435 descent launches of a C3 build: 1.19e6 in total = 1.19 s, 2.7e3 each)
        r["kernel_stats"] = [{"kernel": short(x[0]) or x[0][:60], "calls": x[1], "total_ms": round(x[2] / 1e3, 3), "avg_ms": round(x[3] / 1e3, 4)}
                             for x in rows[:8]]
    pm = defaultdict(lambda: defaultdict(float))
    nd = defaultdict(int)
    for sub in (case + "_pmc1", case + "_pmc2"):
        for f in dbs(sub):
            cur = sqlite3.connect(f).cursor()
            seen = set()
            for k, c, v, d, did in cur.execute("select kernel_name, counter_name, value, duration, dispatch_id from counters_collection"):
                s = short(k)
                if s:
                    pm[s][c] += v
                    if (sub, did) not in seen and c in ("SQ_WAVE_CYCLES", "SQ_INSTS_MFMA"):
                        seen.add((sub, did))
                        pm[s]["duration_ms_" + sub[-4:]] += d / 1e6
                        nd[(s, sub[-4:])] += 1
    r["pmc"] = {k: {c: round(v, 3) for c, v in sorted(cs.items())} for k, cs in pm.items()}
    for k, cs in pm.items():
        d = r["pmc"][k]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs and cs["SQ_BUSY_CYCLES"]:
            # gfx94x formula of MfmaUtil (derived_counters.xml has no gfx950 section): MFMA-busy cycles over busy cycles x SIMDs/SE
            d["mfma_busy_over_sq_busy_x4"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (cs["SQ_BUSY_CYCLES"] * 4.0), 4)
        t = cs.get("duration_ms_pmc1")
        if t and "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
            # share of the chip's matrix-pipe cycles that were busy while the kernel ran (all SIMDs, nominal clock)
            d["mfma_pipe_busy_frac_of_chip"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (t * 1e-3 * CLOCK_GHZ * 1e9 * N_SIMD), 4)
        t2 = cs.get("duration_ms_pmc2")
        if t2 and "SQ_INSTS_VALU_MFMA_MOPS_F32" in cs:
            # MOPS counts 512 flop units (one v_mfma_f32_32x32x2_f32 = 4096 flop = 8 MOPS)
            flops = cs["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0
            d["mfma_TFLOPs"] = round(flops / (t2 * 1e-3) / 1e12, 2)
            d["mfma_frac_of_157TF"] = round(flops / (t2 * 1e-3) / 157.3e12, 4)
    res[case] = r
try:
    res["commit"] = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".build_commit")).read().strip()
except OSError:
    res["commit"] = None
print(json.dumps(res, indent=1))
