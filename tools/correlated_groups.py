#!/usr/bin/env python3
"""Group penalties on the correlated designs of tools/correlated_detail.py (groups of 10 adjacent columns): passes,
time, and the certificate per point; the plain iteration of the same path as the referee."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p, K = 70000, 1200, 20
which = sys.argv[1:] or ["ar1_0.95", "lowrank+noise"]
for name in which:
    rng = np.random.default_rng(0)
    E = rng.standard_normal((n, p))
    if name == "ar1_0.95":
        X = E.copy()
        for j in range(1, p):
            X[:, j] = 0.95 * X[:, j - 1] + np.sqrt(1 - 0.95**2) * E[:, j]
    else:
        X = rng.standard_normal((n, 8)) @ rng.standard_normal((8, p)) * 2.0 + 0.3 * rng.standard_normal((n, p))
    groups = np.repeat(np.arange(p // 10), 10)
    coef = np.zeros(p)
    for g in rng.choice(p // 10, 6, replace=False):
        coef[groups == g] = rng.standard_normal(10) * 2
    y = X @ coef + rng.standard_normal(n) * 2
    with eng.dataset(X, y) as ds:
        ds.set_groups(groups, p // 10)
        g0, _ = ds.gradient(None)
        bmax = float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0))))
        for kind, pts in (("group", [(0.0, a, 0.0) for a in np.geomspace(bmax, 1e-2 * bmax, K)]),
                          ("sparse-group", [(0.3 * a, 0.7 * a, 0.0) for a in np.geomspace(bmax, 1e-2 * bmax, K)])):
            r = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_FRESH_L, max_iter=100000)
            t0 = time.perf_counter(); r = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_FRESH_L, max_iter=100000); dt = time.perf_counter() - t0
            q = ds.solve_path(pts, lanes=4, flags=_engine.FLAG_NO_WORKING_SET | _engine.FLAG_FRESH_L, tol=1e-10, max_iter=400000)
            err = np.max(np.abs(r.betas - q.betas)) / np.max(np.abs(q.betas))
            print(f"{name} {kind}: {dt*1e3:.1f} ms, passes {r.grad_launches} (plain {q.grad_launches}), converged {r.converged}/{q.converged}, "
                  f"err vs plain {err:.1e}, direct {r.ws_direct_steps}, inner {r.ws_inner_iters}, refined {r.ws_refined}, cols {r.ws_columns}, "
                  f"active groups {int(np.sum(np.abs(r.betas[-1]).reshape(-1, 10).sum(1) > 0))}", flush=True)
