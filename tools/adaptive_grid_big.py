"""GridSearchCV(AdaptiveLasso) on a 100 000 x 5 000 host array: 12 alphas x 5 folds, every cell a re-weighting loop -- over X
and from the folds' Grams (solver_options covariance False / "auto" / True)."""
import os, sys, time, json, warnings
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.model_selection import KFold
from sparselm_amd.model import AdaptiveLasso
from sparselm_amd.model_selection import GridSearchCV

n, p = 100_000, 5_000
rng = np.random.default_rng(2)
coef = np.zeros(p); coef[rng.choice(p, 60, replace=False)] = 10.0 * rng.standard_normal(60)
X = rng.standard_normal((n, p))
y = X @ coef + 10.0 * rng.standard_normal(n)
amax = float(np.max(np.abs(X.T @ y)) / n)
grid = {"alpha": list(np.geomspace(0.5 * amax, 2e-3 * amax, 12))}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ref = None
    for cov in (False, False, "auto", True, True):
        t0 = time.perf_counter()
        gs = GridSearchCV(AdaptiveLasso(max_iter=5, solver_options={"covariance": cov}), grid, cv=KFold(5, shuffle=True, random_state=0)).fit(X, y)
        dt = time.perf_counter() - t0
        if ref is None:
            ref = gs
        print(json.dumps({"covariance": cov, "seconds": round(dt, 3), "search_seconds": round(gs.search_time_, 3), "best_alpha": float(gs.best_params_["alpha"]),
                          "scores_vs_first": float(np.max(np.abs(gs.cv_results_["mean_test_score"] - ref.cv_results_["mean_test_score"]))),
                          "nnz": int(np.count_nonzero(gs.best_estimator_.coef_)), "rounds": int(gs.best_estimator_.n_iter_)}), flush=True)
