#!/usr/bin/env python3
"""Small fits: wall time per solve against the polling chunk (check_every)."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.datasets import make_regression
from sparselm_amd import _engine
eng = _engine.get_engine(0)
for n, p, alpha in ((25, 30, 0.1), (100, 80, 1e-3), (400, 100, 0.1), (2000, 200, 1.0), (5000, 2000, 0.5)):
    X, y = make_regression(n_samples=n, n_features=p, n_informative=10, noise=1.0, random_state=0)
    with eng.dataset(X, y) as ds:
        out = []
        for ce in (0, 4, 8, 16):
            for _ in range(3):
                ds.solve_path([(alpha, 0, 0)], max_iter=200000, check_every=ce)
            t0 = time.perf_counter()
            for _ in range(20):
                r = ds.solve_path([(alpha, 0, 0)], max_iter=200000, check_every=ce)
            dt = (time.perf_counter() - t0) / 20
            out.append(f"chunk={ce}: {1e3*dt:6.3f} ms")
        print(f"n={n} p={p}: {int(r.n_iter[0])} passes | " + " | ".join(out), flush=True)
