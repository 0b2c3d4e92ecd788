#!/usr/bin/env python3
"""BASELINE config 1 (README example: AdaptiveLasso in GridSearchCV on make_regression(100 x 80)) and a larger
adaptive grid: device path (loops side by side as lanes) against scikit-learn's loop of fits on the same engine."""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.datasets import make_regression
from sklearn.model_selection import GridSearchCV as SkGridSearchCV
from sparselm_amd.model import AdaptiveLasso
from sparselm_amd.model_selection import GridSearchCV
warnings.simplefilter("ignore")
for n, p, grid in [(100, 80, np.logspace(-8, 2, 10)), (20000, 2000, np.geomspace(30, 0.3, 10))]:
    X, y = make_regression(n_samples=n, n_features=p, n_informative=10, noise=0.0 if n == 100 else 5.0, random_state=0)
    est = AdaptiveLasso(fit_intercept=False)
    for rep in range(2):
        t = time.perf_counter(); fast = GridSearchCV(est, {"alpha": list(grid)}).fit(X, y); tf = time.perf_counter() - t
    t = time.perf_counter(); slow = SkGridSearchCV(est, {"alpha": list(grid)}, scoring="neg_root_mean_squared_error").fit(X, y); ts = time.perf_counter() - t
    print(f"n={n} p={p}: device path {tf*1e3:.1f} ms, generic loop {ts*1e3:.1f} ms; best alpha {fast.best_params_['alpha']:.3g} / {slow.best_params_['alpha']:.3g};"
          f" best score {fast.best_score_:.6g} / {slow.best_score_:.6g}; nnz {np.count_nonzero(fast.best_estimator_.coef_)} / {np.count_nonzero(slow.best_estimator_.coef_)}", flush=True)
