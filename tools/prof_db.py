#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 results database (rocpd sqlite): calls, total, average, share."""
import glob, os, sqlite3, sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
dbs = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True), key=os.path.getmtime)
if not dbs:
    sys.exit(f"no .db under {root}")
cur = sqlite3.connect(dbs[-1]).cursor()
rows = cur.execute(
    'select name, count(*), sum("end"-start), avg("end"-start), min("end"-start), max("end"-start) '
    "from kernels group by name order by 3 desc"
).fetchall()
tot = sum(r[2] for r in rows)
print(f"# {dbs[-1]}")
print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'share':>6s}")
for name, n, t, avg, mn, mx in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{name[:72]:72s} {n:6d} {t/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:10.2f} {100*t/tot:5.1f}%")
