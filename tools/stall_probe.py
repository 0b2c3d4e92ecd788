#!/usr/bin/env python3
"""Per-call wall times of repeated identical solves (looks for sporadic host-side stalls)."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.datasets import make_regression
from sparselm_amd import _engine

eng = _engine.get_engine(0)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for n, p in ((2000, 1000),):
    X, y = make_regression(n_samples=n, n_features=p, n_informative=10, noise=1.0, random_state=0)
    amax = np.max(np.abs(X.T @ y)) / n
    pts = [(a, 0, 0) for a in np.geomspace(amax, 1e-3 * amax, 20)]
    with eng.dataset(X, y) as ds:
        for name, fl in (("plain", _engine.FLAG_NO_WORKING_SET), ("ws", _engine.FLAG_WORKING_SET)):
            ts = []
            for _ in range(REPS):
                t0 = time.perf_counter()
                r = ds.solve_path(pts, max_iter=200000, flags=fl)
                ts.append(1e3 * (time.perf_counter() - t0))
            slow = [(i, round(t, 1)) for i, t in enumerate(ts) if t > 10]
            print(n, p, name, "passes", r.grad_launches, f"median {np.median(ts):.2f} ms mean {np.mean(ts):.2f} ms stalls {slow}", flush=True)
