"""AdaptiveGroupLasso on the reference's 25 x 30 fixture size with SLM_TRACE=2: what each of its solves costs."""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.datasets import make_regression
from sparselm_amd.model import AdaptiveGroupLasso, GroupLasso
warnings.simplefilter("ignore")
X, y = make_regression(n_samples=25, n_features=30, n_informative=10, random_state=1)
groups = np.arange(30) // 5
AdaptiveGroupLasso(groups=groups, alpha=0.1, fit_intercept=True).fit(X, y)
os.environ["SLM_TRACE"] = "2"
t0 = time.perf_counter()
m = AdaptiveGroupLasso(groups=groups, alpha=0.1, fit_intercept=True).fit(X, y)
print(f"AdaptiveGroupLasso.fit: {1e3 * (time.perf_counter() - t0):.2f} ms, n_iter_ {m.n_iter_}", file=sys.stderr)
t0 = time.perf_counter()
m = GroupLasso(groups=groups, alpha=0.1, fit_intercept=True).fit(X, y)
print(f"GroupLasso.fit: {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr)
