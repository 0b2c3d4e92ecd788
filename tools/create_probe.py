#!/usr/bin/env python3
"""Cost of creating / destroying a dataset and of a whole estimator fit on small problems."""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
from sparselm_amd.model import Lasso
eng = _engine.get_engine(0)
rng = np.random.default_rng(0)
for n, p in ((25, 30), (400, 100), (2000, 200), (5000, 2000)):
    X = rng.standard_normal((n, p)); y = rng.standard_normal(n)
    for _ in range(3):
        eng.dataset(X, y).close()
    t0 = time.perf_counter()
    for _ in range(20):
        ds = eng.dataset(X, y)
        ds.close()
    t_create = (time.perf_counter() - t0) / 20
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Lasso(alpha=0.1).fit(X, y)
        t0 = time.perf_counter()
        for _ in range(20):
            Lasso(alpha=0.1).fit(X, y)
        t_fit = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20):
            Lasso(alpha=0.1, fit_intercept=True).fit(X, y)
        t_fit_i = (time.perf_counter() - t0) / 20
    print(f"n={n} p={p}: dataset create+destroy {1e3*t_create:.3f} ms | Lasso.fit {1e3*t_fit:.3f} ms | with intercept {1e3*t_fit_i:.3f} ms", flush=True)
