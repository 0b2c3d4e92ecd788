"""SparseGroupLasso(standardize=True) at the README's size with SLM_TRACE=2: sweeps, products, b-steps solved directly."""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.datasets import make_regression
from sparselm_amd.model import SparseGroupLasso
warnings.simplefilter("ignore")
X, y = make_regression(n_samples=100, n_features=80, n_informative=10, random_state=0)
SparseGroupLasso(groups=np.arange(80) // 8, alpha=0.5, standardize=True, fit_intercept=True).fit(X, y)
os.environ["SLM_TRACE"] = "2"
for alpha in (0.5, 0.05, 5.0):
    t0 = time.perf_counter()
    m = SparseGroupLasso(groups=np.arange(80) // 8, alpha=alpha, standardize=True, fit_intercept=True).fit(X, y)
    print(f"alpha {alpha}: {1e3 * (time.perf_counter() - t0):.2f} ms, {m.solver_info_['n_iter']} sweeps, "
          f"{np.count_nonzero(m.coef_)} non-zeros", file=sys.stderr)
