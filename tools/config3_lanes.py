#!/usr/bin/env python3
"""BASELINE config 3's path (GroupLasso, 500 shuffled groups of 10, 50 alphas) on several lane counts: ms and passes per path.
usage: config3_lanes.py [lanes ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
import bench
from sparselm_amd import _engine
eng = _engine.get_engine(0)
c4 = bench.Config4(eng, 100_000, 5_000)
ds, G, groups, K = c4.ds, c4.G, c4.groups, c4.K
g0, _ = ds.gradient(None)
bmax = float(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G)).max())
alphas = np.geomspace(bmax, 1e-3 * bmax, K)
pts = np.c_[0 * alphas, alphas, 0 * alphas]
ref = None
for lanes in [int(a) for a in sys.argv[1:]] or (16, 17, 18, 20, 25, 32, 0):
    for _ in range(2):
        r = ds.solve_path(pts, tol=1e-8, lanes=lanes, flags=_engine.FLAG_FRESH_L)
    eng.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        r = ds.solve_path(pts, tol=1e-8, lanes=lanes, flags=_engine.FLAG_FRESH_L)
    eng.synchronize(); dt = (time.perf_counter() - t0) / 5
    if ref is None: ref = r.betas.copy()
    err = float(np.max(np.abs(r.betas - ref)) / np.max(np.abs(ref)))
    print(f"config 3 lanes={lanes}: {1e3*dt:.3f} ms per path = {K/dt:.0f} fits/s, {r.grad_launches} passes, ws b/a/m/cols {r.ws_builds}/{r.ws_appends}/{r.ws_misses}/{r.ws_columns}, conv {r.converged}, vs first {err:.1e}", flush=True)
c4.close()
