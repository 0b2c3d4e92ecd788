#!/usr/bin/env python3
"""Tall X (n = 1e6, p = 1000; 8 GB): 16-lane working-set path against the plain 4-lane iteration."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_003, int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.default_rng(0)
coef = np.zeros(p); coef[rng.choice(p, 40, replace=False)] = 10 * rng.standard_normal(40)
with eng.synthetic_dataset(n, p, seed=3, coef=coef, noise_sd=5.0) as ds:
    g0, _ = ds.gradient(None)
    amax = float(np.max(np.abs(g0)))
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, 40)]
    for rep in range(2):
        t = time.perf_counter(); r = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_FRESH_L); dt = time.perf_counter() - t
    t = time.perf_counter(); q = ds.solve_path(pts, lanes=4, flags=_engine.FLAG_NO_WORKING_SET, tol=1e-9); dq = time.perf_counter() - t
    err = float(np.max(np.abs(r.betas - q.betas)) / np.max(np.abs(q.betas)))
    print(f"n={n} p={p}: working set {dt*1e3:.1f} ms / {r.grad_launches} passes (cols {r.ws_columns}); plain {dq*1e3:.1f} ms / {q.grad_launches} passes; rel-inf {err:.1e}; converged {r.converged} {q.converged}")
