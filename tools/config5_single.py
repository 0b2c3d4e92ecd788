#!/usr/bin/env python3
"""BASELINE config 5 on ONE rank's share: AdaptiveGroupLasso (3 re-weighting solves) on 125000 x 10000
with a single-rank RCCL communicator (the all-reduces run, over one rank)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine, distributed as D
n, p, G = 125_000, 10_000, 1_000
groups = np.repeat(np.arange(G), 10)
rng = np.random.default_rng(0)
coef = np.zeros(p)
for g in rng.choice(G, 30, replace=False):
    coef[groups == g] = rng.uniform(1, 5, 10)
eng = _engine.Engine(0)
D.init_row_sharding(eng, rank=0, world_size=1)
ds = eng.synthetic_dataset(n, p, seed=1000, coef=coef, noise_sd=5.0)
ds.set_global_rows(n)
ds.set_groups(groups, G)
g0, _, _ = ds.gradient(None, reps=10)
amax = float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G))))
alpha, eps = 0.1 * amax, 1e-6
for name, fl in (("working set", 0), ("plain", _engine.FLAG_NO_WORKING_SET)):
    for rep in range(2):
        w = alpha * np.ones(G); beta = None; passes = 0
        t0 = time.perf_counter()
        for _ in range(3):
            res = ds.solve_path([(0.0, 1.0, 0.0)], b=w, beta0=beta, tol=1e-8, want_group_norms=True, flags=fl)
            beta = res.betas[0]; passes += res.grad_launches
            w = alpha * (alpha / (res.group_norms[0] + eps))
        dt = time.perf_counter() - t0
    print(json.dumps({"mode": name, "seconds_3_solves": round(dt, 4), "passes": passes, "converged": res.converged,
                      "active_groups": int(np.sum(res.group_norms[0] > 0)), "ws": [res.ws_builds, res.ws_appends, res.ws_refined]}), flush=True)
    if name == "working set": b_ws = beta.copy()
print("rel-inf diff working set vs plain:", float(np.max(np.abs(b_ws - beta)) / np.max(np.abs(beta))))
ds.close(); eng.comm_destroy(); eng.close()
