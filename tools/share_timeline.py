"""Condense a rocprofv3 kernel trace (csv) of tools/config4_share_trace.py: the LAST solve in it (from its last
`solve_setup_kernel` on): per-kernel totals, launches and the idle time between kernels.
usage: python tools/share_timeline.py <kernel_trace.csv> [list]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("slm::", "").replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# a solve opens with power_init_kernel (the step-size seed) or solve_setup_kernel
opens = [i for i, s in enumerate(seq) if s[0].startswith("power_init")] or [i for i, s in enumerate(seq) if s[0].startswith("solve_setup")]
start = opens[-1]
t0 = seq[start][1]
tot, cnt, gaps, prev = {}, {}, 0.0, None
for name, a, b in seq[start:]:
    if prev is not None and a > prev:
        gaps += (a - prev) / 1e3
        if len(sys.argv) > 2 and (a - prev) / 1e3 > 8:
            print(f"{(a - t0) / 1e3:9.1f} us  idle {(a - prev) / 1e3:7.1f} us before {name[:50]}")
    tot[name] = tot.get(name, 0.0) + (b - a) / 1e3
    cnt[name] = cnt.get(name, 0) + 1
    prev = max(prev or b, b)
print(f"window {(seq[-1][2] - t0) / 1e3:.1f} us, idle between kernels {gaps:.1f} us")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:24]:
    print(f"{v:9.1f} us  {cnt[k]:4d} x {v / cnt[k]:8.1f}  {k[:70]}")
