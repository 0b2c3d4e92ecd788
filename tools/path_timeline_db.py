#!/usr/bin/env python3
"""Timeline of the LAST path solve in a rocprofv3 results database (rocpd sqlite): every kernel with its start
offset, the gap to the previous kernel's end and its duration, then totals per kernel and the sum of the gaps.

usage: python tools/path_timeline_db.py <dir with *.db> [passes of the path, default 5]"""
import glob, os, sqlite3, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 5
db = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True), key=os.path.getmtime)[-1]
cur = sqlite3.connect(db).cursor()
rows = cur.execute('select name, start, "end" from kernels order by start').fetchall()
seq = [(n.split("(")[0].replace("slm::", "").replace("void ", ""), a, b) for n, a, b in rows]
xs = [i for i, s in enumerate(seq) if s[0].startswith("xtr_mfma") and s[2] - s[1] > 100000]
first = xs[-passes]
# the solve starts at the power_init_kernel before its first working pass
start = max(i for i in range(first) if seq[i][0].startswith("power_init"))
end = max(i for i, s in enumerate(seq) if s[0].startswith("ws_solve") and s[2] - s[1] > 8000)
t0 = seq[start][1]
prev = None
tot, gaps = {}, 0.0
for name, a, b in seq[start:end + 1]:
    gap = (a - prev) / 1e3 if prev else 0.0
    dur = (b - a) / 1e3
    tot[name] = tot.get(name, [0.0, 0])
    tot[name][0] += dur
    tot[name][1] += 1
    gaps += gap
    print(f"{(a - t0) / 1e3:9.1f} us  gap {gap:6.1f}  {dur:8.1f} us  {name[:60]}")
    prev = b
print("--- totals over the solve (us)")
for k, (v, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{v:9.1f}  x{c:<3d} {k}")
print(f"kernels {sum(v for v, _ in tot.values()):.1f} us + gaps {gaps:.1f} us = window {(seq[end][2] - t0) / 1e3:.1f} us")
