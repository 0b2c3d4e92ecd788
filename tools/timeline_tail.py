"""Timeline of the last `count` launches of a rocprofv3 kernel trace (csv): start offset, gap to the predecessor's end,
duration, name.  usage: python tools/timeline_tail.py <kernel_trace.csv> [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("slm::", "").replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
seq = seq[-(int(sys.argv[2]) if len(sys.argv) > 2 else 80):]
t0, prev = seq[0][1], None
for name, a, b in seq:
    print(f"{(a - t0) / 1e3:9.1f} us  gap {((a - prev) / 1e3 if prev else 0.0):7.1f}  {(b - a) / 1e3:8.1f} us  {name[:70]}")
    prev = b
