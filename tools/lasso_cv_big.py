"""GridSearchCV(Lasso) over 50 alphas x 5 folds on a 100 000 x 5 000 host array (the LassoCV workflow at scale): seconds per
search, over X and from the folds' Grams, and what scikit-learn's LassoCV takes on the host cores for the same grid."""
import os, sys, time, json, warnings
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.model_selection import KFold
from sparselm_amd.model import Lasso
from sparselm_amd.model_selection import GridSearchCV

n, p = 100_000, 5_000
rng = np.random.default_rng(2)
coef = np.zeros(p); coef[rng.choice(p, 60, replace=False)] = 10.0 * rng.standard_normal(60)
X = rng.standard_normal((n, p))
y = X @ coef + 10.0 * rng.standard_normal(n)
amax = float(np.max(np.abs(X.T @ y)) / n)
grid = {"alpha": list(np.geomspace(amax, 1e-2 * amax, 50))}
cv = KFold(5, shuffle=True, random_state=0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for cov in (False, False, "auto", True, True):
        t0 = time.perf_counter()
        gs = GridSearchCV(Lasso(solver_options={"covariance": cov}), grid, cv=cv).fit(X, y)
        dt = time.perf_counter() - t0
        print(json.dumps({"covariance": cov, "seconds": round(dt, 4), "search_seconds": round(gs.search_time_, 4), "best_alpha_over_max": float(gs.best_params_["alpha"]) / amax,
                          "nnz": int(np.count_nonzero(gs.best_estimator_.coef_))}), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "sklearn":
    from sklearn.linear_model import LassoCV
    t0 = time.perf_counter()
    m = LassoCV(alphas=grid["alpha"], cv=cv, fit_intercept=False, n_jobs=16).fit(X, y)
    print(json.dumps({"sklearn_LassoCV_seconds": round(time.perf_counter() - t0, 2), "best_alpha_over_max": float(m.alpha_) / amax}))
