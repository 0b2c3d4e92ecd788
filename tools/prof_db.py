#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 results database (rocpd sqlite): calls, total, average, share."""
import glob, os, sqlite3, sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
dbs = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True), key=os.path.getmtime)
if not dbs:
    sys.exit(f"no .db under {root}")
cur = sqlite3.connect(dbs[-1]).cursor()
# "work" = launches longer than a tenth of the kernel's longest one: the queue runs up to two chunks
# ahead of the device-side stop flag, and those launches return at once (4 us)
rows = cur.execute(
    'select k.name, count(*), sum(k."end"-k.start), avg(k."end"-k.start), min(k."end"-k.start), max(k."end"-k.start), '
    'sum(case when (k."end"-k.start) * 10 >= m.mx then 1 else 0 end), '
    'sum(case when (k."end"-k.start) * 10 >= m.mx then (k."end"-k.start) else 0 end) '
    'from kernels k join (select name, max("end"-start) as mx from kernels group by name) m on m.name = k.name '
    "group by k.name order by 3 desc"
).fetchall()
tot = sum(r[2] for r in rows)
print(f"# {dbs[-1]}")
print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'share':>6s} {'work':>6s} {'work_avg_us':>12s}")
for name, n, t, avg, mn, mx, nw, tw in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{name[:72]:72s} {n:6d} {t/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:10.2f} {100*t/tot:5.1f}% {nw:6d} {tw/max(nw,1)/1e3:12.2f}")
