#!/usr/bin/env python3
"""Where the wall time of one headline path goes on the host side: Python wrapper vs C entry (SLM_TRACE=2 prints
the C side's own marks)."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p = 100000, 5000
rng = np.random.default_rng(0)
coef = np.zeros(p); coef[rng.choice(p, 50, replace=False)] = 100 * rng.uniform(size=50)
ds = eng.synthetic_dataset(n, p, seed=7, coef=coef, noise_sd=10.0)
g0, _ = ds.gradient(None)
amax = float(np.max(np.abs(g0)))
pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, 50)]
for _ in range(2):
    ds.solve_path(pts, lanes=16, flags=_engine.FLAG_FRESH_L)
for rep in range(8):
    ce = 0 if rep % 2 == 0 else 2  # check_every=2: fixed chunks of two passes (the schedule before the expected-pass cut)
    t = time.perf_counter()
    r = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_FRESH_L, check_every=ce)
    dt = (time.perf_counter() - t) * 1e3
    print(f"check_every={ce} python wall {dt:.3f} ms   C wall {r.wall_ms:.3f} ms   lipschitz {r.lipschitz_ms:.3f} ms  passes {r.grad_launches}", flush=True)
