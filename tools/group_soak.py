#!/usr/bin/env python3
"""The group-lasso path of BASELINE config 3's shape (n = 100k, p = 5k, 500 groups of 10, 50 alphas, the engine's choice of lanes: contiguous ranges
with work stealing, the sample start) on many random datasets: converged, pass count, agreement with the plain four-lane
iteration of the same data.  usage: group_soak.py [seeds]"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p, G = 100000, 5000, 500
groups = np.repeat(np.arange(G), p // G)
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
LANES = int(os.environ.get("GROUP_SOAK_LANES", "0"))  # (0: the engine's choice -- twenty-five for fifty points in contiguous ranges)
worst, bad, passes = 0.0, 0, []
for seed in range(seeds):
    rng = np.random.default_rng(500 + seed)
    k = int(rng.integers(3, 60))                       # informative groups
    noise = float(rng.choice([0.1, 10.0, 100.0]))
    lo = float(rng.choice([0.1, 0.03, 0.01]))
    coef = np.zeros(p)
    for g in rng.choice(G, k, replace=False):
        coef[groups == g] = 10.0 * rng.standard_normal(p // G) * (rng.random(p // G) < rng.choice([1.0, 0.5]))
    perm = rng.permutation(p)                           # groups not contiguous in the columns
    with eng.synthetic_dataset(n, p, seed=700 + seed, coef=coef[np.argsort(perm)], noise_sd=noise) as ds:
        gid = groups[np.argsort(perm)]
        ds.set_groups(gid, G)
        g0, _ = ds.gradient(None)
        bmax = float(np.max(np.sqrt(np.bincount(gid, weights=g0 * g0, minlength=G))))
        pts = [(0.0, a, 0.0) for a in np.geomspace(bmax, lo * bmax, 50)]
        ds.solve_path(pts, lanes=LANES)
        t = time.perf_counter(); r = ds.solve_path(pts, lanes=LANES); dt = (time.perf_counter() - t) * 1e3
        q = ds.solve_path(pts, lanes=4, flags=_engine.FLAG_NO_WORKING_SET, tol=1e-9)
        err = float(np.max(np.abs(r.betas - q.betas)) / max(np.max(np.abs(q.betas)), 1e-300))
        act = int(np.count_nonzero(np.bincount(gid, weights=np.abs(r.betas[-1]), minlength=G)))
        worst = max(worst, err); passes.append(int(r.grad_launches))
        flag = "" if (r.converged and q.converged and err < 1e-6) else "  <-- CHECK"
        bad += bool(flag)
        print(f"seed {seed:2d} groups={k:2d} noise={noise:5.1f} lo={lo:4.2f}: {dt:6.2f} ms, {r.grad_launches:2d} passes (plain: {q.grad_launches}), ws b/a/m/cols {r.ws_builds}/{r.ws_appends}/{r.ws_misses}/{r.ws_columns}, active groups {act}, err {err:.1e}{flag}", flush=True)
print(f"GROUP SOAK seeds {seeds} worst rel-inf {worst:.2e} flagged {bad} passes min/median/max {min(passes)}/{int(np.median(passes))}/{max(passes)}")
