#!/usr/bin/env python3
"""BASELINE config 3 alone (GroupLasso 500x10, 50-alpha path, 16 lanes) for profiling and knob sweeps."""
import os, sys, time, json
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
n, p, G = 100_000, 5_000, 500
rng = np.random.default_rng(1)
groups = rng.permutation(np.repeat(np.arange(G), 10))
coef = np.zeros(p)
for g in rng.choice(G, 25, replace=False):
    coef[groups == g] = 100.0 * rng.uniform(size=10)
eng = _engine.get_engine(0)
ds = eng.synthetic_dataset(n, p, seed=11, coef=coef, noise_sd=10.0)
ds.set_groups(groups, G)
g0, _, _ = ds.gradient(None, reps=30)
gnorm = np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G))
bmax = float(gnorm.max())
pts = [(0.0, a, 0.0) for a in np.geomspace(bmax, 1e-3 * bmax, 50)]
import itertools
grid = dict(SLM_WS_APPEND=["16", "64", "160"], SLM_WS_KINIT=["112", "256", "384"], SLM_WS_LOOKAHEAD=["2", "6"])
if len(sys.argv) > 1:
    grid = dict(SLM_WS_APPEND=[sys.argv[1]], SLM_WS_KINIT=[sys.argv[2]], SLM_WS_LOOKAHEAD=[sys.argv[3]])
for combo in itertools.product(*grid.values()):
    for k, v in zip(grid.keys(), combo):
        os.environ[k] = v
    ds.solve_path(pts, lanes=16)
    t0 = time.perf_counter()
    for _ in range(3):
        res = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_FRESH_L)
    dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"knobs": combo, "ms_per_path": round(1e3 * dt, 2), "passes": res.grad_launches, "ws": [res.ws_builds, res.ws_appends, res.ws_refined, res.ws_misses, res.ws_columns], "max_iter": int(max(res.n_iter))}), flush=True)
