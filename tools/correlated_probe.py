#!/usr/bin/env python3
"""Large correlated designs (AR(1) columns, and low rank + noise): the sketched L seed (3 power steps on a
sixteenth of the rows) and the working set against the plain iteration."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p = 70000, 1200
rng = np.random.default_rng(0)
for name in ("ar1_0.95", "lowrank+noise", "scaled_columns"):
    E = rng.standard_normal((n, p))
    if name == "ar1_0.95":
        X = E.copy()
        for j in range(1, p):
            X[:, j] = 0.95 * X[:, j - 1] + np.sqrt(1 - 0.95**2) * E[:, j]
    elif name == "lowrank+noise":
        X = rng.standard_normal((n, 8)) @ rng.standard_normal((8, p)) * 2.0 + 0.3 * E
    else:
        X = E * rng.uniform(0.01, 30.0, p)
    coef = np.zeros(p); coef[rng.choice(p, 25, replace=False)] = rng.standard_normal(25) * 3
    y = X @ coef + rng.standard_normal(n) * 2
    with eng.dataset(X, y) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, 30)]
        t = time.perf_counter(); r = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_FRESH_L, max_iter=200000); dt = time.perf_counter() - t
        t = time.perf_counter(); q = ds.solve_path(pts, lanes=4, flags=_engine.FLAG_NO_WORKING_SET | _engine.FLAG_FRESH_L, tol=1e-9, max_iter=200000); dq = time.perf_counter() - t
        err = float(np.max(np.abs(r.betas - q.betas)) / np.max(np.abs(q.betas)))
        print(f"{name:15s}: working set {dt*1e3:8.1f} ms / {r.grad_launches:5d} passes (cols {r.ws_columns}, misses {r.ws_misses}); plain {dq*1e3:9.1f} ms / {q.grad_launches:6d} passes; rel-inf {err:.1e}; converged {r.converged} {q.converged}", flush=True)
