#!/usr/bin/env python3
"""The headline path on 16 ... 32 lanes: passes and time per path (a lane count above sixteen runs two halves on one read of X)."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import make_coef
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p, K = 100000, 5000, 50
coef = make_coef(p, 50, seed=0)
with eng.synthetic_dataset(n, p, seed=1000, coef=coef, noise_sd=10.0) as ds:
    g0, _ = ds.gradient(None)
    amax = float(np.max(np.abs(g0)))
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
    ref = None
    for lanes in [int(a) for a in sys.argv[1:]] or (16, 17, 18, 20, 25, 32, 16, 17):
        for _ in range(3):
            ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L)
        eng.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            r = ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L)
        eng.synchronize(); dt = (time.perf_counter() - t0) / 10
        if ref is None: ref = r.betas.copy()
        err = float(np.max(np.abs(r.betas - ref)) / np.max(np.abs(ref)))
        print(f"headline lanes={lanes}: {1e3*dt:.3f} ms per path = {K/dt:.0f} fits/s, {r.grad_launches} passes, ws b/a/m/cols {r.ws_builds}/{r.ws_appends}/{r.ws_misses}/{r.ws_columns}, conv {r.converged}, vs first {err:.1e}", flush=True)
