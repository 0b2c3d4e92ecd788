"""Covariance passes for ONE row set at several widths (p = 530, 1 530, 5 000): Gram build time and ms per path against passes over X."""
import os, sys, time
sys.path.insert(0, "/root/repo/sparse-lm_amd"); sys.path.insert(0, "/root/repo")
import numpy as np
from sparselm_amd import _engine
eng = _engine.get_engine(0)
for p in (530, 1530, 5000):
    coef = np.zeros(p); coef[:10] = 1.0
    ds = eng.synthetic_dataset(100_000, p, seed=3, coef=coef, noise_sd=1.0)
    eng.synchronize()
    for rep in range(3):
        ds.covariance_clear()
        eng.synchronize()
        t0 = time.perf_counter(); ds.covariance(None, 0); eng.synchronize(); dt = time.perf_counter() - t0
        print(f"p = {p}: covariance(None, 0) {1e3 * dt:.2f} ms", flush=True)
    ds.close()
