"""cProfile of the README example (BASELINE config 1: GridSearchCV(AdaptiveLasso) on make_regression(100, 80)) and of a
reference-sized Lasso.fit: where the host time of the small-problem path goes."""
import cProfile, os, pstats, sys, time, warnings
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.datasets import make_regression
from sparselm_amd.model import AdaptiveLasso, Lasso
from sparselm_amd.model_selection import GridSearchCV
warnings.simplefilter("ignore")
X, y = make_regression(n_samples=100, n_features=80, n_informative=10, random_state=0)
grid = {"alpha": np.logspace(-8, 2, 10)}
GridSearchCV(AdaptiveLasso(fit_intercept=False), grid).fit(X, y)
for _ in range(3):
    t0 = time.perf_counter()
    gs = GridSearchCV(AdaptiveLasso(fit_intercept=False), grid).fit(X, y)
    print(f"README grid: {1e3 * (time.perf_counter() - t0):.2f} ms, best alpha {gs.best_params_['alpha']:.3g}")
pr = cProfile.Profile()
pr.enable()
GridSearchCV(AdaptiveLasso(fit_intercept=False), grid).fit(X, y)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
Xs, ys = make_regression(n_samples=25, n_features=30, n_informative=10, random_state=1)
Lasso(alpha=0.1).fit(Xs, ys)
t0 = time.perf_counter()
for _ in range(50):
    Lasso(alpha=0.1).fit(Xs, ys)
print(f"Lasso.fit 25x30, cached dataset: {1e3 * (time.perf_counter() - t0) / 50:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    Lasso(alpha=0.1).fit(Xs, ys)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
