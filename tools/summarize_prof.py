#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into small tracked files under profiles/.

Usage: python tools/summarize_prof.py <round-tag>   (e.g. r01a)
Reads  gpurun_out/prof_stats/**/_kernel_stats.csv      (rocprofv3 --kernel-trace --stats)
       gpurun_out/prof_fetch/**/_counter_collection.csv (rocprofv3 --pmc FETCH_SIZE, own pass)
       gpurun_out/prof_write/**/_counter_collection.csv (rocprofv3 --pmc WRITE_SIZE, own pass)
Writes profiles/<tag>_kernel_stats.csv, profiles/<tag>_hbm_counters.json
HBM byte conventions follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): counters are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming
read, so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def counters(tag):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "**", "*_counter_collection.csv"), recursive=True)
    out = {}
    if not files:
        return out
    per = {}
    files.sort(key=os.path.getmtime)
    for r in csv.DictReader(open(files[-1])):  # newest run
        per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        # launches that early-exit (path already finished) move ~0 bytes: keep the working ones
        big = [x for x in v if x > 0.5 * max(v)] or v  # (a kernel that never moved a byte: all of its launches)
        out[k] = {"launches": len(v), "working_launches": len(big), "median_kib": statistics.median(big), "max_kib": max(v)}
    return out


def main():
    tag = sys.argv[1]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    stats = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_stats", "**", "*_kernel_stats.csv"), recursive=True)
    stats.sort(key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    fetch, write = counters("fetch"), counters("write")
    summary = {"units": "KiB per dispatch as reported by rocprofv3; *_bytes fields are corrected bytes", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        e = {}
        if k in fetch:
            e["FETCH_SIZE"] = fetch[k]
            e["read_bytes_corrected"] = fetch[k]["median_kib"] * 1024 * 2
        if k in write:
            e["WRITE_SIZE"] = write[k]
            e["write_bytes"] = write[k]["median_kib"] * 1024
        if "read_bytes_corrected" in e and "write_bytes" in e:
            e["hbm_bytes_per_launch"] = e["read_bytes_corrected"] + e["write_bytes"]
        summary["kernels"][k] = e
    path = os.path.join(ROOT, "profiles", f"{tag}_hbm_counters.json")
    json.dump(summary, open(path, "w"), indent=1)
    print("wrote", path)
    for k, e in summary["kernels"].items():
        if "grad_fused" in k:
            print(k, json.dumps(e, indent=1))


if __name__ == "__main__":
    main()
