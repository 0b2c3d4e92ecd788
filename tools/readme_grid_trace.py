import os, sys, time, warnings
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.datasets import make_regression
from sparselm_amd.model import AdaptiveLasso
from sparselm_amd.model_selection import GridSearchCV
warnings.simplefilter("ignore")
X, y = make_regression(n_samples=100, n_features=80, n_informative=10, random_state=0)
def run():
    s = GridSearchCV(AdaptiveLasso(fit_intercept=False), {"alpha": np.logspace(-8, 2, 10)})
    s.fit(X, y); return s
for _ in range(5): run()
t0 = time.perf_counter()
for _ in range(20): s = run()
print("README grid ms", 1e3 * (time.perf_counter() - t0) / 20, s.best_params_)
os.environ["SLM_TRACE"] = "2"
run()
