#!/usr/bin/env python3
"""Sweep the working-set knobs on the headline path (env vars read by the engine per solve)."""
import itertools, os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
sys.path.insert(0, ROOT)
from sparselm_amd import _engine
from bench import make_coef

eng = _engine.get_engine(0)
n, p, K = 100_000, 5_000, 50
ds = eng.synthetic_dataset(n, p, seed=1000, coef=make_coef(p, 50, seed=0), noise_sd=10.0)
g0, _, _ = ds.gradient(None, reps=50)
amax = float(np.max(np.abs(g0)))
pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
grid = dict(SLM_WS_THETA=["0.85", "0.95"], SLM_WS_LOOKAHEAD=["2"], SLM_WS_APPEND=["24", "48", "96"], SLM_WS_KINIT=["64", "112"])
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    grid = dict(SLM_WS_THETA=["0.7"], SLM_WS_LOOKAHEAD=["4"], SLM_WS_APPEND=["32"], SLM_WS_KINIT=["112"])
for combo in itertools.product(*grid.values()):
    for k, v in zip(grid.keys(), combo):
        os.environ[k] = v
    ds.solve_path(pts, tol=1e-8, lanes=10, flags=_engine.FLAG_FRESH_L)
    t0 = time.perf_counter()
    for _ in range(4):
        r = ds.solve_path(pts, tol=1e-8, lanes=10, flags=_engine.FLAG_FRESH_L)
    dt = (time.perf_counter() - t0) / 4
    print(dict(zip(grid.keys(), combo)), f"{1e3*dt:7.2f} ms/path {K/dt:7.0f} fits/s passes={r.grad_launches} builds={r.ws_builds} "
          f"appends={r.ws_appends} misses={r.ws_misses} cols={r.ws_columns} conv={r.converged}", flush=True)
