#!/usr/bin/env python3
"""Timeline of the last solve in a rocprofv3 kernel trace (csv): start offset, gap to the previous kernel, duration.

usage: python tools/path_timeline.py <kernel_trace.csv> [n_xtr_launches_back [shortest launch listed, us]]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
xs = [i for i, s in enumerate(seq) if ("xtr" in s[0] and "_mfma_kernel" in s[0]) or "grad_ring" in s[0]]  # (xtr_, xtr18_, xtr20_, xtr32_mfma_kernel)
back = int(sys.argv[2]) if len(sys.argv) > 2 else 9
start = max(0, xs[-back] - 45)
t0 = seq[start][1]
prev = None
tot = {}
for name, a, b in seq[start:]:
    nm = name.split("(")[0].replace("slm::", "").replace("void ", "")
    gap = (a - prev) / 1e3 if prev else 0.0
    dur = (b - a) / 1e3
    tot[nm] = tot.get(nm, 0.0) + dur
    if dur > float(sys.argv[3] if len(sys.argv) > 3 else 12) or gap > 12:
        print(f"{(a - t0) / 1e3:9.1f} us  gap {gap:6.1f}  {dur:8.1f} us  {nm[:60]}")
    prev = b
print("--- totals over the window (us)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:16]:
    print(f"{v:9.1f}  {k}")
print(f"window {(seq[-1][2] - t0) / 1e3:.1f} us")
