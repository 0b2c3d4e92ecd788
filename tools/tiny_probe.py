#!/usr/bin/env python3
"""Passes / wall time of single solves on a tiny, nearly unregularised problem (README example sizes)."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.datasets import make_regression
from sparselm_amd import _engine
eng = _engine.get_engine(0)
X, y = make_regression(n_samples=100, n_features=80, n_informative=10, random_state=0)
rng = np.random.default_rng(0)
mask = np.ones(100); mask[rng.choice(100, 20, replace=False)] = 0
flagsets = {"default": 0, "ws": _engine.FLAG_WORKING_SET, "plain": _engine.FLAG_NO_WORKING_SET}
with eng.dataset(X, y) as ds:
    for alpha in (1e2, 1.0, 1e-2, 1e-4, 1e-8):
        for name, fl in flagsets.items():
            for lanes in (1, 4):
                specs = [dict(points=[(alpha, 0, 0)], row_weight=mask, n_eff=80) for _ in range(lanes)]
                ds.solve_lanes(specs, flags=fl)
                t = time.perf_counter(); R = ds.solve_lanes(specs, flags=fl); dt = time.perf_counter() - t
                r = R[0]
                print(f"alpha={alpha:7.0e} {name:8s} lanes={lanes:2d} passes={r.grad_launches:6d} n_iter={int(r.n_iter[0]):6d} conv={r.converged} wall={dt*1e3:8.2f} ms ws(b/a/r/m)={r.ws_builds}/{r.ws_appends}/{r.ws_refined}/{r.ws_misses}", flush=True)
