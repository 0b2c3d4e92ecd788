import os, sys, time, warnings
import numpy as np
sys.path.insert(0, "/root/repo/sparse-lm_amd")
from sklearn.datasets import make_regression
from sparselm_amd.model import AdaptiveGroupLasso
warnings.simplefilter("ignore")
X, y = make_regression(n_samples=25, n_features=30, n_informative=10, random_state=1)
groups = np.arange(30) // 5
AdaptiveGroupLasso(groups=groups, alpha=0.1, fit_intercept=True).fit(X, y)
os.environ["SLM_TRACE"] = "2"
t0 = time.perf_counter()
m = AdaptiveGroupLasso(groups=groups, alpha=0.1, fit_intercept=True).fit(X, y)
print("fit ms", 1e3 * (time.perf_counter() - t0), "rounds", m.n_iter_, file=sys.stderr)
