#!/usr/bin/env python3
"""cProfile of BASELINE config 4 through the scikit-learn surface (GridSearchCV(SparseGroupLasso), 2 500 fits on a 100 000 x 5 000
host array): where the HOST spends what the engine does not -- the third search of a process (dataset cached, library warm)."""
import cProfile, os, pstats, sys, time, warnings
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
X = rng.standard_normal((n, p))
y = X @ coef + 10.0 * rng.standard_normal(n)
c = X.T @ y / n
bmax = float(np.max(np.sqrt(np.bincount(groups, weights=c * c, minlength=G))))
grid = {"alpha": list(np.geomspace(bmax, 1e-3 * bmax, 50)), "l1_ratio": list(np.linspace(0.05, 0.95, 10))}
warnings.simplefilter("ignore")
def search():
    return GridSearchCV(SparseGroupLasso(groups=groups), grid, cv=KFold(5, shuffle=True, random_state=0)).fit(X, y)
search(); search()
pr = cProfile.Profile()
t0 = time.perf_counter(); pr.enable(); gs = search(); pr.disable(); dt = time.perf_counter() - t0
print(f"third search {dt:.3f} s (search_time_ {gs.search_time_:.3f})")
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
