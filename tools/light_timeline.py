#!/usr/bin/env python3
"""A few headline paths on a draw whose last point misses (data seed 7): run under `rocprofv3 --kernel-trace` to see the
kernels of a certified partial pass (light_kernels.hpp) in the timeline.  usage: light_timeline.py [data seed]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
import numpy as np
from bench import make_coef
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p, K = 100000, 5000, 50
dseed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
with eng.synthetic_dataset(n, p, seed=dseed, coef=make_coef(p, 50, seed=0), noise_sd=10.0) as ds:
    g0, _ = ds.gradient(None)
    amax = float(np.max(np.abs(g0)))
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
    for _ in range(6):
        r = ds.solve_path(pts, lanes=0, flags=_engine.FLAG_FRESH_L)
    print(f"seed {dseed}: passes {r.grad_launches} light {r.light_passes} columns {r.light_columns} wall {r.wall_ms:.3f} ms")
