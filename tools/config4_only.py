#!/usr/bin/env python3
"""BASELINE config 4 alone (engine level, ten (fold, l1_ratio) units per pass) for profiling."""
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
l1_ratios = np.linspace(0.05, 0.95, 10)
folds = np.random.default_rng(0).permutation(n) % 5
masks = [(folds != f).astype(float) for f in range(5)]
units = [(f, r) for f in range(5) for r in l1_ratios]
def run_grid():
    total = 0
    for k0 in range(0, len(units), 10):
        specs = []
        for f, r in units[k0:k0 + 10]:
            amax = min(bmax / (1 - r), float(np.max(np.abs(g0))) / r)
            al = np.geomspace(amax, 1e-3 * amax, 50)
            specs.append(dict(points=np.c_[r * al, (1 - r) * al, 0 * al], row_weight=masks[f], n_eff=int(masks[f].sum())))
        out = ds.solve_lanes(specs)
        total += out[0].grad_launches
        assert all(o.converged for o in out)
    return total, out[0]
run_grid()
t0 = time.perf_counter(); passes, r = run_grid(); dt = time.perf_counter() - t0
print(json.dumps({"seconds": dt, "passes": passes, "ws": [r.ws_builds, r.ws_appends, r.ws_refined, r.ws_misses, r.ws_columns]}))
