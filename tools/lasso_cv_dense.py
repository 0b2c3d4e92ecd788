"""GridSearchCV(Lasso) whose grid reaches into the dense regime (noise-fitting alphas: solutions of thousands of non-zeros) on a
100 000 x 5 000 host array: seconds per search and where they go (cProfile of one search)."""
import cProfile, os, pstats, sys, time, json, warnings
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
y = X @ coef + 100.0 * rng.standard_normal(n)
amax = float(np.max(np.abs(X.T @ y)) / n)
grid = {"alpha": list(np.geomspace(amax, 1e-3 * amax, 50))}
cv = KFold(5, shuffle=True, random_state=0)
warnings.simplefilter("ignore")
for cov in (False, "auto", True):
    for rep in range(2):
        t0 = time.perf_counter()
        gs = GridSearchCV(Lasso(solver_options={"covariance": cov}), grid, cv=cv).fit(X, y)
        print(json.dumps({"covariance": cov, "seconds": round(time.perf_counter() - t0, 3), "search_seconds": round(gs.search_time_, 3),
                          "best_alpha_over_max": float(gs.best_params_["alpha"]) / amax, "nnz": int(np.count_nonzero(gs.best_estimator_.coef_))}), flush=True)
pr = cProfile.Profile(); pr.enable()
GridSearchCV(Lasso(solver_options={"covariance": True}), grid, cv=cv).fit(X, y)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
