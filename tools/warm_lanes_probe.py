#!/usr/bin/env python3
"""Cost of warm starts in a multi-lane call on large X: every lane with a beta0 takes its first residual from X
(rowdot), the cold ones from -y."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p = 100000, 5000
rng = np.random.default_rng(0)
coef = np.zeros(p); coef[rng.choice(p, 50, replace=False)] = 100 * rng.uniform(size=50)
ds = eng.synthetic_dataset(n, p, seed=7, coef=coef, noise_sd=10.0)
g0, _ = ds.gradient(None)
amax = float(np.max(np.abs(g0)))
alphas = np.geomspace(0.5 * amax, 0.01 * amax, 16)
cold = [dict(points=[(a, 0.0, 0.0)]) for a in alphas]
R = ds.solve_lanes(cold)
for rep in range(3):
    t = time.perf_counter(); R = ds.solve_lanes(cold); tc = time.perf_counter() - t
    warm = [dict(points=[(0.9 * a, 0.0, 0.0)], beta0=r.betas[0]) for a, r in zip(alphas, R)]
    t = time.perf_counter(); W = ds.solve_lanes(warm); tw = time.perf_counter() - t
    print(f"16 lanes x 1 point: cold {tc*1e3:.2f} ms ({R[0].grad_launches} passes), warm-started at 0.9 alpha {tw*1e3:.2f} ms ({W[0].grad_launches} passes)", flush=True)
