#!/usr/bin/env python3
"""One line per bench.py log: fits/s, ms per path, gradient-kernel ms, passes, everything but the passes.

usage: python tools/benchline.py <bench log> [...]"""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads([ln for ln in open(path).read().splitlines() if ln.startswith("{")][-1])  # (rocprofv3 logs after it)
    r = d["roofline"]
    passes = d["config"]["grad_evals_per_path"]
    print(f"{path}: {d['value']:.0f} {d['unit']}, {d['ms_per_step']:.3f} ms/path, kernel {r['avg_kernel_ms']:.4f} ms "
          f"({100 * r['frac']:.1f} %), {passes:g} passes, outside the passes {d['ms_per_step'] - passes * r['avg_kernel_ms']:.3f} ms, "
          f"path-level {100 * r['path_level']['frac']:.1f} %")
