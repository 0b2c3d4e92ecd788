"""Condense a `rocprofv3 --pmc A B ...` run (csv) into profiles/<name>.json: per kernel and counter the median over the
working launches (those above half of the kernel's largest value of its first counter), and -- when SQ_VALU_MFMA_BUSY_CYCLES
and GRBM_GUI_ACTIVE are among them -- the matrix pipe's share of the SIMD cycles (GRBM_GUI_ACTIVE is summed over the 8 XCDs:
cycles = sum / 8; 1 024 SIMDs).  usage: python tools/summarize_counters.py <dir under gpurun_out> <name> "<the command that was run>" """
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
src, name, run = sys.argv[1], sys.argv[2], sys.argv[3]
files = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", src, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
per = {}
for r in csv.DictReader(open(files[-1])):
    per.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {"run": run, "kernels": {}}
for k, cs in sorted(per.items()):
    first = next(iter(cs.values()))
    keep = [i for i, v in enumerate(first) if v > 0.5 * max(first)] or list(range(len(first)))
    e = {"launches": len(first), "working_launches": len(keep)}
    for c, v in cs.items():
        e[c + "_median"] = statistics.median([v[i] for i in keep if i < len(v)])
    if "SQ_VALU_MFMA_BUSY_CYCLES_median" in e and e.get("GRBM_GUI_ACTIVE_median", 0) > 0:
        e["mfma_busy_fraction_of_simd_cycles"] = e["SQ_VALU_MFMA_BUSY_CYCLES_median"] / (e["GRBM_GUI_ACTIVE_median"] / 8.0 * 1024.0)
    out["kernels"][k] = e
path = os.path.join(ROOT, "profiles", name + ".json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path)
