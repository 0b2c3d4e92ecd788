"""The on-chip kernel's own account of a reference-sized `Lasso.fit` (25 x 30, alpha = 0.1): SLM_TRACE=2 lines of three fits."""
import os, sys, warnings
os.environ["SLM_TRACE"] = "2"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sparse-lm_amd"))
from sklearn.datasets import make_regression
from sparselm_amd.model import Lasso
warnings.simplefilter("ignore")
Xs, ys = make_regression(n_samples=25, n_features=30, n_informative=10, random_state=1)
for _ in range(3):
    m = Lasso(alpha=0.1).fit(Xs, ys)
print("nnz", int((m.coef_ != 0).sum()), m.solver_info_)
