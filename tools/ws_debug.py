#!/usr/bin/env python3
"""Per-point iteration counts with / without the working-set refinement on a small hard problem."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine

eng = _engine.get_engine(0)
rng = np.random.default_rng(6)
n, p = 60, 200
Z = rng.standard_normal((n, 8))
X = Z @ rng.standard_normal((8, p)) + 0.05 * rng.standard_normal((n, p))
y = X[:, :5] @ np.ones(5) + 0.01 * rng.standard_normal(n)
c = X.T @ y / n
amax = np.max(np.abs(c))
alphas = np.geomspace(amax, 1e-2 * amax, 6)
pts = [(a, 0, 0) for a in alphas]
with eng.dataset(X, y) as ds:
    for name, fl in (("ws", _engine.FLAG_WORKING_SET), ("plain", _engine.FLAG_NO_WORKING_SET)):
        r = ds.solve_path(pts, tol=1e-10, max_iter=int(sys.argv[1]) if len(sys.argv) > 1 else 30000, flags=fl)
        print(name, "n_iter", r.n_iter, "status", r.status, "mode", r.mode, "resid", r.resid, "bnorm", r.beta_norm,
              "nnz", [int(np.count_nonzero(b)) for b in r.betas], "ws", r.ws_builds, r.ws_refined, r.ws_misses, flush=True)
