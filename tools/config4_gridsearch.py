#!/usr/bin/env python3
"""BASELINE config 4 end to end through the scikit-learn surface: GridSearchCV(SparseGroupLasso) with
50 alphas x 10 l1_ratios x 5 folds on a 100000 x 5000 host array (one GPU)."""
import os, sys, time, json, warnings
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.model_selection import KFold
from sparselm_amd.model import SparseGroupLasso
from sparselm_amd.model_selection import GridSearchCV

n, p, G = 100_000, 5_000, 500
rng = np.random.default_rng(1)
groups = rng.permutation(np.repeat(np.arange(G), 10))
coef = np.zeros(p)
for g in rng.choice(G, 25, replace=False):
    coef[groups == g] = 100.0 * rng.uniform(size=10)
t0 = time.perf_counter()
X = rng.standard_normal((n, p))
y = X @ coef + 10.0 * rng.standard_normal(n)
print(f"host data {time.perf_counter()-t0:.1f} s", flush=True)
c = X.T @ y / n
bmax = float(np.max(np.sqrt(np.bincount(groups, weights=c * c, minlength=G))))
grid = {"alpha": list(np.geomspace(bmax, 1e-3 * bmax, 50)), "l1_ratio": list(np.linspace(0.05, 0.95, 10))}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep, streams in enumerate((1, 1, 2, 3)):
        t0 = time.perf_counter()
        gs = GridSearchCV(SparseGroupLasso(groups=groups), grid, cv=KFold(5, shuffle=True, random_state=0), streams=streams).fit(X, y)
        dt = time.perf_counter() - t0
        print(json.dumps({"rep": rep, "streams": streams, "seconds": round(dt, 3), "search_seconds": round(gs.search_time_, 3), "fits": 2500, "fits_per_s": round(2500 / dt, 1),
                          "best": {k: float(v) for k, v in gs.best_params_.items()}, "best_score": float(gs.best_score_),
                          "nnz_groups": int(np.sum(np.bincount(groups, weights=gs.best_estimator_.coef_ != 0) > 0))}), flush=True)
    # the same search from the folds' Grams (solver_options covariance=True): the Grams' build is part of the search time
    for rep, streams in enumerate((1, 1, 3)):
        t0 = time.perf_counter()
        gs2 = GridSearchCV(SparseGroupLasso(groups=groups, solver_options={"covariance": True}), grid,
                           cv=KFold(5, shuffle=True, random_state=0), streams=streams).fit(X, y)
        dt = time.perf_counter() - t0
        print(json.dumps({"covariance": True, "rep": rep, "streams": streams, "seconds": round(dt, 3), "search_seconds": round(gs2.search_time_, 3),
                          "best": {k: float(v) for k, v in gs2.best_params_.items()},
                          "scores_vs_over_x": float(np.max(np.abs(gs2.cv_results_["mean_test_score"] - gs.cv_results_["mean_test_score"])))}), flush=True)
    # a line search over the two parameters (reference model_selection.py:427-707): four lines on ONE device dataset
    from sparselm_amd.model_selection import LineSearchCV
    t0 = time.perf_counter()
    ls = LineSearchCV(SparseGroupLasso(groups=groups), [("alpha", grid["alpha"]), ("l1_ratio", grid["l1_ratio"])],
                      cv=KFold(5, shuffle=True, random_state=0), n_iter=4).fit(X, y)
    print(json.dumps({"line_search": True, "lines": 4, "seconds": round(time.perf_counter() - t0, 3),
                      "line_search_seconds": [round(h.search_time_, 3) for h in ls.history_], "best": {k: float(v) for k, v in ls.best_params_.items()}}), flush=True)
