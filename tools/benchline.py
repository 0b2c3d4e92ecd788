#!/usr/bin/env python3
"""One line per bench log: file, ms per path, fits/s, the hot kernel's average launch (tools/ab_env.sh prints its runs with it)."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{f}: {d['ms_per_step']:.3f} ms per path, {d['value']:.0f} fits/s, kernel {d['roofline']['avg_kernel_ms']:.4f} ms, {d['roofline']['launches']} launches")
    except Exception as exc:  # noqa: BLE001
        print(f"{f}: unreadable ({exc!r})")
