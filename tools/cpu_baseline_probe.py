#!/usr/bin/env python3
"""Per-iteration cost of the C oracle on the host at the headline size (diagnostics for bench.py's cpu_baseline)."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
import oracle
from oracle import cref
from sparselm_amd import _engine
n, p = 100000, 5000
eng = _engine.get_engine(0)
coef = np.zeros(p); coef[:50] = np.linspace(1, 100, 50)
ds = eng.synthetic_dataset(n, p, seed=1000, coef=coef, noise_sd=10.0)
X0, y = ds.download(); ds.close()
print("threads", cref.num_threads(), "OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"), flush=True)
gidx, G = oracle.group_index(None, p)
with cref.NumaMatrix(X0) as X:
    z = np.random.default_rng(0).standard_normal(p)
    for _ in range(3):
        t = time.perf_counter(); cref.gradient(X, y, z); print("isolated gradient ms", round(1e3 * (time.perf_counter() - t), 1), flush=True)
    amax = np.max(np.abs(X0.T @ y)) / n if False else 40.0
    for alpha in (20.0, 5.0):
        t = time.perf_counter(); b, it = cref.fista(X, y, alpha, 0.0, 0.0, gidx, G, L=1.6, tol=1e-8, max_iter=200); dt = time.perf_counter() - t
        print(f"fista alpha={alpha}: {abs(it)} iterations, {1e3*dt/abs(it):.1f} ms per iteration", flush=True)
