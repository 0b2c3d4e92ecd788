"""A few on-chip solves of reference-sized problems, for `rocprofv3 --kernel-trace` / SLM_TRACE=2."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.datasets import make_regression
from sparselm_amd import _engine
eng = _engine.get_engine(0)
for n, p, alpha in ((25, 30, 0.1), (100, 80, 1e-3), (400, 100, 0.1)):
    X, y = make_regression(n_samples=n, n_features=p, n_informative=10, noise=1.0, random_state=0)
    with eng.dataset(X, y) as ds:
        for _ in range(4):
            t0 = time.perf_counter()
            r = ds.solve_path([(alpha, 0, 0)], max_iter=20000, flags=_engine.FLAG_ON_CHIP)
            print(f"n={n} p={p}: {1e3 * (time.perf_counter() - t0):.3f} ms, mode {r.mode[0]}, {int(r.n_iter[0])} sweeps/passes", file=sys.stderr)
