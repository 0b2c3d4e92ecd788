#!/usr/bin/env python3
"""Reference-sized problems: wall time per solve with and without the working-set refinement."""
import os, sys, time, warnings
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.datasets import make_regression
from sparselm_amd import _engine

eng = _engine.get_engine(0)
cases = [(25, 30, 0.1), (100, 80, 1e-3), (100, 80, 1e-8), (400, 100, 0.1), (400, 300, 0.05), (2000, 200, 1.0), (2000, 1000, 0.5),
         (5000, 2000, 0.5), (20000, 2000, 0.5)]
for n, p, alpha in cases:
    X, y = make_regression(n_samples=n, n_features=p, n_informative=10, noise=1.0, random_state=0)
    with eng.dataset(X, y) as ds:
        out = []
        for name, fl in (("plain", _engine.FLAG_NO_WORKING_SET), ("ws", _engine.FLAG_WORKING_SET)):
            ds.solve_path([(alpha, 0, 0)], max_iter=200000, flags=fl)
            t0 = time.perf_counter()
            for _ in range(5):
                r = ds.solve_path([(alpha, 0, 0)], max_iter=200000, flags=fl)
            dt = (time.perf_counter() - t0) / 5
            out.append(f"{name}: {1e3*dt:8.3f} ms, {int(r.n_iter[0]):6d} passes, conv={r.converged}")
        # a 20-point path
        amax = np.max(np.abs(X.T @ y)) / n
        pts = [(a, 0, 0) for a in np.geomspace(amax, 1e-3 * amax, 20)]
        for name, fl in (("plain", _engine.FLAG_NO_WORKING_SET), ("ws", _engine.FLAG_WORKING_SET)):
            ds.solve_path(pts, max_iter=200000, flags=fl)
            t0 = time.perf_counter()
            for _ in range(3):
                r = ds.solve_path(pts, max_iter=200000, flags=fl)
            dt = (time.perf_counter() - t0) / 3
            out.append(f"path20 {name}: {1e3*dt:8.3f} ms, {int(r.grad_launches):6d} passes")
    print(f"n={n} p={p} alpha={alpha}: " + " | ".join(out), flush=True)
