import cProfile, pstats, os, sys, time, warnings
import numpy as np
sys.path.insert(0, "/root/repo/sparse-lm_amd")
from sklearn.model_selection import KFold
from sparselm_amd.model import SparseGroupLasso
from sparselm_amd.model_selection import GridSearchCV
n, p, G = 100_000, 5_000, 500
rng = np.random.default_rng(1)
groups = rng.permutation(np.repeat(np.arange(G), 10))
coef = np.zeros(p)
for g in rng.choice(G, 25, replace=False):
    coef[groups == g] = 100.0 * rng.uniform(size=10)
X = rng.standard_normal((n, p)); y = X @ coef + 10.0 * rng.standard_normal(n)
c = X.T @ y / n
bmax = float(np.max(np.sqrt(np.bincount(groups, weights=c * c, minlength=G))))
grid = {"alpha": list(np.geomspace(bmax, 1e-3 * bmax, 50)), "l1_ratio": list(np.linspace(0.05, 0.95, 10))}
warnings.simplefilter("ignore")
for _ in range(2):
    GridSearchCV(SparseGroupLasso(groups=groups), grid, cv=KFold(5, shuffle=True, random_state=0)).fit(X, y)
pr = cProfile.Profile(); pr.enable()
GridSearchCV(SparseGroupLasso(groups=groups), grid, cv=KFold(5, shuffle=True, random_state=0)).fit(X, y)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
