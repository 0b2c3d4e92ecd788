"""Per-kernel table of a rocprofv3 kernel trace (csv): calls, total, average, min, max -- optionally only the launches
after the N-th last occurrence of a marker kernel.  usage: python tools/kernel_table.py <kernel_trace.csv> [marker] [n-th last]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("slm::", "").replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
if len(sys.argv) > 2:
    hits = [i for i, s in enumerate(seq) if s[0].startswith(sys.argv[2])]
    seq = seq[hits[-(int(sys.argv[3]) if len(sys.argv) > 3 else 1)]:]
tot = {}
for name, a, b in seq:
    t = tot.setdefault(name, [0, 0.0, 1e30, 0.0])
    d = (b - a) / 1e3
    t[0] += 1; t[1] += d; t[2] = min(t[2], d); t[3] = max(t[3], d)
busy = sum(v[1] for v in tot.values())
print(f"window {(seq[-1][2] - seq[0][1]) / 1e3:.1f} us, kernels {busy:.1f} us")
print(f"{'kernel':60s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{k[:60]:60s} {v[0]:6d} {v[1]:10.1f} {v[1] / v[0]:9.1f} {v[2]:9.1f} {v[3]:9.1f}")
