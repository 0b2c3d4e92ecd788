#!/usr/bin/env python3
"""Wall time of reference-sized fits (launch-bound regime): engine solves and whole estimator fits."""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.datasets import make_regression
from sparselm_amd import _engine
from sparselm_amd.model import Lasso, AdaptiveGroupLasso

eng = _engine.get_engine(0)
for n, p, alpha in ((25, 30, 0.1), (100, 80, 1e-3), (400, 100, 0.1), (2000, 200, 1.0)):
    X, y = make_regression(n_samples=n, n_features=p, n_informative=10, noise=1.0, random_state=0)
    for what, flags in (("general path", 0), ("on-chip", _engine.FLAG_ON_CHIP)):
        with eng.dataset(X, y) as ds:
            ds.solve_path([(alpha, 0, 0)], max_iter=20000, flags=flags)
            t0 = time.perf_counter()
            for _ in range(5):
                r = ds.solve_path([(alpha, 0, 0)], max_iter=20000, flags=flags)
            dt = (time.perf_counter() - t0) / 5
        print(f"n={n} p={p} alpha={alpha} {what}: {1e3*dt:8.3f} ms per solve, {int(r.n_iter[0])} {'sweeps' if r.mode[0] == 2 else 'passes'}, "
              f"converged={r.converged}", flush=True)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    X, y = make_regression(n_samples=25, n_features=30, n_informative=10, random_state=1)
    Lasso(alpha=0.1).fit(X, y)
    t0 = time.perf_counter()
    for k in range(10):
        Lasso(alpha=0.1).fit(X * (1.0 + 1e-3 * k), y)  # (new content every time: no dataset-cache hit)
    print(f"Lasso.fit 25x30 (incl. upload/alloc): {1e2*(time.perf_counter()-t0):.3f} ms per fit")
    t0 = time.perf_counter()
    for k in range(10):
        Lasso(alpha=0.1).fit(X, y)
    print(f"Lasso.fit 25x30 on data the dataset cache holds: {1e2*(time.perf_counter()-t0):.3f} ms per fit")
    groups = np.arange(30) // 5
    t0 = time.perf_counter()
    for _ in range(10):
        AdaptiveGroupLasso(groups=groups, alpha=0.1, fit_intercept=True).fit(X, y)
    print(f"AdaptiveGroupLasso.fit 25x30 (3 re-weighting solves): {1e2*(time.perf_counter()-t0):.2f} ms per fit")
    from sklearn.model_selection import KFold
    from sparselm_amd.model import AdaptiveLasso, SparseGroupLasso
    from sparselm_amd.model_selection import GridSearchCV
    Xr, yr = make_regression(n_samples=100, n_features=80, n_informative=10, random_state=0)
    GridSearchCV(AdaptiveLasso(fit_intercept=False), {"alpha": np.logspace(-8, 2, 10)}).fit(Xr, yr)
    t0 = time.perf_counter()
    GridSearchCV(AdaptiveLasso(fit_intercept=False), {"alpha": np.logspace(-8, 2, 10)}).fit(Xr, yr)
    print(f"README example (BASELINE config 1: GridSearchCV(AdaptiveLasso), 10 alphas x 5 folds + refit, 100x80): "
          f"{1e3*(time.perf_counter()-t0):.1f} ms")
    t0 = time.perf_counter()
    m = SparseGroupLasso(groups=np.arange(80) // 8, alpha=0.5, standardize=True, fit_intercept=True).fit(Xr, yr)
    print(f"SparseGroupLasso(standardize=True).fit 100x80: {1e3*(time.perf_counter()-t0):.1f} ms, "
          f"{m.solver_info_['n_iter']} sweeps, {m.solver_info_['inner_iterations']} gradient evaluations")
