#!/usr/bin/env python3
"""How long does scikit-learn's coordinate-descent lasso_path take on the headline workload (host cores)?"""
import os, sys, time
os.environ.setdefault("OPENBLAS_NUM_THREADS", sys.argv[1] if len(sys.argv) > 1 else "64")
os.environ.setdefault("OMP_NUM_THREADS", os.environ["OPENBLAS_NUM_THREADS"])
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd")); sys.path.insert(0, ROOT)
from sparselm_amd import _engine
from bench import make_coef
from sklearn.linear_model import lasso_path
eng = _engine.get_engine(0)
n, p, K = 100_000, 5_000, 50
ds = eng.synthetic_dataset(n, p, seed=1000, coef=make_coef(p, 50, seed=0), noise_sd=10.0)
g0, _ = ds.gradient(None)
amax = float(np.max(np.abs(g0)))
alphas = np.geomspace(amax, 1e-3 * amax, K)
t0 = time.perf_counter(); X, y = ds.download(); print(f"download {time.perf_counter()-t0:.2f} s", flush=True)
res = ds.solve_path([(a, 0, 0) for a in alphas], tol=1e-8, lanes=10)
Xf = np.asfortranarray(X); print("fortran copy done", flush=True)
for pre in (True,):
    t0 = time.perf_counter()
    al, coefs, gaps = lasso_path(Xf, y, alphas=alphas, precompute=pre, tol=1e-10, max_iter=100000)
    dt = time.perf_counter() - t0
    err = np.max(np.abs(coefs.T - res.betas)) / np.max(np.abs(res.betas))
    print(f"lasso_path precompute={pre}: {dt:.2f} s -> {K/dt:.2f} fits/s, threads {os.environ['OPENBLAS_NUM_THREADS']}, max dual gap {gaps.max():.2e}, rel-inf diff vs GPU {err:.2e}", flush=True)
