"""One `Lasso.fit` on a 100 000 x 5 000 host array (4 GB): seconds, and where they go (cProfile)."""
import cProfile, os, pstats, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sparse-lm_amd"))
from sparselm_amd import model
warnings.simplefilter("ignore")
rng = np.random.default_rng(4)
n, p = 100_000, 5_000
X = rng.standard_normal((n, p)); y = X[:, :50] @ (rng.standard_normal(50) * 10) + 10.0 * rng.standard_normal(n) + 1.5
amax = float(np.max(np.abs(X.T @ (y - y.mean()) / n)))
for fi in (False, True):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); m = model.Lasso(alpha=0.05 * amax, fit_intercept=fi).fit(X, y); ts.append(time.perf_counter() - t0)
    print(f"fit_intercept={fi}: {ts[0]:.3f} s first, {min(ts[1:]):.3f} s then; nnz {int(np.count_nonzero(m.coef_))}", flush=True)
pr = cProfile.Profile(); pr.enable()
model.Lasso(alpha=0.05 * amax, fit_intercept=True).fit(X, y)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
