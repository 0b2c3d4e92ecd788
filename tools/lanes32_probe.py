#!/usr/bin/env python3
"""The headline path on 16 and on 32 lanes (two halves on one read of X), and the pass kernels on their own.
usage: lanes32_probe.py [seed ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import make_coef, soak_case
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p, K = 100000, 5000, 50
coef = make_coef(p, 50, seed=0)
with eng.synthetic_dataset(n, p, seed=1000, coef=coef, noise_sd=10.0) as ds:
    g0, _ = ds.gradient(None)
    amax = float(np.max(np.abs(g0)))
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
    ref = None
    for lanes in (16, 32, 16, 32):
        for _ in range(3):
            ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L)
        eng.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            r = ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L)
        eng.synchronize(); dt = (time.perf_counter() - t0) / 10
        if ref is None: ref = r.betas.copy()
        err = float(np.max(np.abs(r.betas - ref)) / np.max(np.abs(ref)))
        print(f"headline lanes={lanes}: {1e3*dt:.3f} ms per path = {K/dt:.0f} fits/s, {r.grad_launches} passes, ws b/a/m/cols {r.ws_builds}/{r.ws_appends}/{r.ws_misses}/{r.ws_columns}, conv {r.converged}, vs first {err:.1e}", flush=True)
    os.environ["SLM_GRAD_SPLIT"] = "1"; os.environ["SLM_GRAD_SPLIT_XTR_ONLY"] = "1"
    for B in (16, 32):
        os.environ["SLM_PROBE_LANES"] = str(B)
        _, _, ms = ds.gradient(None, reps=20)
        print(f"X^T R alone, {B} lanes: {ms:.4f} ms")
    del os.environ["SLM_GRAD_SPLIT_XTR_ONLY"]
    for B in (16, 32):
        os.environ["SLM_PROBE_LANES"] = str(B)
        _, _, ms = ds.gradient(None, reps=20)
        print(f"residuals from X + X^T R, {B} lanes: {ms:.4f} ms")
    for k in ("SLM_GRAD_SPLIT", "SLM_PROBE_LANES"): del os.environ[k]
for seed in [int(a) for a in sys.argv[1:]]:
    coef, noise, lo, k = soak_case(seed, p)
    with eng.synthetic_dataset(n, p, seed=100 + seed, coef=coef, noise_sd=noise) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, lo * amax, K)]
        out = []
        for lanes in (16, 32):
            ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L)
            t0 = time.perf_counter(); r = ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L); dt = time.perf_counter() - t0
            out.append((lanes, 1e3 * dt, r.grad_launches, r.converged, r.betas))
        err = float(np.max(np.abs(out[0][4] - out[1][4])) / np.max(np.abs(out[0][4])))
        print(f"soak seed {seed}: " + "; ".join(f"lanes={l}: {ms:.2f} ms / {pa} passes conv {c}" for l, ms, pa, c, _ in out) + f"; 16 vs 32 {err:.1e}, nnz_last {np.count_nonzero(out[0][4][-1])}", flush=True)
