#!/usr/bin/env python3
"""Per-point detail of the working-set path on the two correlated designs of tests/test_direct_solve_gpu.py
(error against scikit-learn, passes, KKT residual, mu, direct steps)."""
import os, sys, warnings
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.linear_model import lasso_path
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p, K = 70000, 1200, 30
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
    coef = np.zeros(p); coef[rng.choice(p, 25, replace=False)] = rng.standard_normal(25) * 3
    y = X @ coef + rng.standard_normal(n) * 2
    with eng.dataset(X, y) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        alphas = np.geomspace(amax, 1e-3 * amax, K)
        r = ds.solve_path([(a, 0.0, 0.0) for a in alphas], lanes=16, flags=_engine.FLAG_FRESH_L)
        import time
        t0 = time.perf_counter(); r = ds.solve_path([(a, 0.0, 0.0) for a in alphas], lanes=16, flags=_engine.FLAG_FRESH_L); dt = time.perf_counter() - t0
        print(f"{name}: second solve {dt * 1e3:.2f} ms")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ref, _ = lasso_path(X, y, alphas=alphas, precompute=X.T @ X, Xy=X.T @ y, tol=1e-14, max_iter=400000)
    ref = ref.T
    scale = np.max(np.abs(ref))
    err = np.max(np.abs(r.betas - ref), axis=1) / scale
    print(f"{name}: passes {r.grad_launches}, err max {err.max():.2e}, direct steps {r.ws_direct_steps}, inner iters {r.ws_inner_iters}, "
          f"refined {r.ws_refined}, misses {r.ws_misses}, builds {r.ws_builds}, appends {r.ws_appends}, cols {r.ws_columns}, converged {r.converged}")
    for k in range(K):
        print(f"  pt {k:2d} nnz {int(np.sum(r.betas[k] != 0)):4d}/{int(np.sum(ref[k] != 0)):4d} n_iter {r.n_iter[k]:3d} mode {r.mode[k]} err {err[k]:.1e} "
              f"kkt {r.kkt[k]:.1e} mu {r.mu[k]:.2e} kkt/mu/|b| {r.kkt[k] / max(r.mu[k], 1e-300) / max(r.beta_norm[k], 1e-300):.1e}")
