"""cProfile of the README example (GridSearchCV over AdaptiveLasso, 100 x 80): where the host spends the time of the reference-sized case."""
import cProfile, os, pstats, sys, time, warnings
import numpy as np
sys.path.insert(0, "/root/repo/sparse-lm_amd")
from sklearn.datasets import make_regression
from sparselm_amd.model import AdaptiveLasso
from sparselm_amd.model_selection import GridSearchCV
warnings.simplefilter("ignore")
X, y = make_regression(n_samples=100, n_features=80, n_informative=10, random_state=0)
grid = {"alpha": np.logspace(-8, 2, 10)}
for _ in range(3):
    GridSearchCV(AdaptiveLasso(fit_intercept=False), grid).fit(X, y)
ts=[]
for _ in range(10):
    t0=time.perf_counter(); GridSearchCV(AdaptiveLasso(fit_intercept=False), grid).fit(X, y); ts.append(time.perf_counter()-t0)
print("median ms", 1e3*np.median(ts))
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    GridSearchCV(AdaptiveLasso(fit_intercept=False), grid).fit(X, y)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(32)
