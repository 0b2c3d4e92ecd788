"""Durations (us) of every launch of the kernels whose name contains PATTERN, in launch order, from a rocprofv3 kernel trace csv.
usage: kernel_durations.py <kernel_trace.csv> PATTERN [PATTERN ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for pat in sys.argv[2:]:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]]
    print(pat, len(d), "launches:", " ".join(f"{x:.0f}" for x in d))
