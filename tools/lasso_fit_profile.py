"""Where a reference-sized `Lasso.fit` (25 x 30) spends its time on the host: wall time and the cProfile table."""
import cProfile, os, pstats, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sparse-lm_amd"))
from sklearn.datasets import make_regression
from sparselm_amd.model import Lasso
warnings.simplefilter("ignore")
Xs, ys = make_regression(n_samples=25, n_features=30, n_informative=10, random_state=1)
for _ in range(20):
    Lasso(alpha=0.1).fit(Xs, ys)
t0 = time.perf_counter()
for _ in range(200):
    Lasso(alpha=0.1).fit(Xs, ys)
print("ms per fit", 1e3 * (time.perf_counter() - t0) / 200)
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    Lasso(alpha=0.1).fit(Xs, ys)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
