#!/usr/bin/env python3
"""Extra measurements for BASELINE.json configs 3 and 4 on ONE GPU (not the headline bench line).

config 3: GroupLasso, 500 groups x 10 (shuffled labels), n=100k p=5k, 50-alpha path.
config 4: SparseGroupLasso, 5 folds x 10 l1_ratios x 50 alphas = 2500 fits: every (fold, l1_ratio)
          unit is one warm-started path; folds are row masks; four units share each pass over X.
Prints one JSON line per config.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--p", type=int, default=5_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--lanes", type=int, default=_engine.MAX_LANES)
    args = ap.parse_args()
    n, p = args.n, args.p
    G = p // 10
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

    # ---- config 3 -----------------------------------------------------------------------------
    bmax = float(gnorm.max())
    alphas = np.geomspace(bmax, 1e-3 * bmax, 50)
    pts = [(0.0, a, 0.0) for a in alphas]
    ds.solve_path(pts, lanes=args.lanes)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        res = ds.solve_path(pts, lanes=args.lanes, flags=_engine.FLAG_FRESH_L)
    dt = (time.perf_counter() - t0) / args.reps
    print(json.dumps({"config": f"3: GroupLasso 500x10 groups, 50-alpha path, 1 GPU, {args.lanes} lanes", "fits_per_s": 50 / dt,
                      "ms_per_path": 1e3 * dt, "passes": res.grad_launches, "converged": res.converged, "ws": [res.ws_builds, res.ws_appends, res.ws_refined, res.ws_misses, res.ws_columns],
                      "nnz_groups_last": int(np.sum(res.betas[-1].reshape(-1) != 0) // 10)}), flush=True)

    # ---- config 4 -----------------------------------------------------------------------------
    l1_ratios = np.linspace(0.05, 0.95, 10)
    folds = np.random.default_rng(0).permutation(n) % 5  # KFold(5, shuffle=True)
    masks = [(folds != f).astype(float) for f in range(5)]
    units = [(f, r) for f in range(5) for r in l1_ratios]  # fold-major: a batch shares one row mask, one Gram

    def run_grid():
        total_passes = 0
        for k0 in range(0, len(units), args.lanes):
            specs = []
            batch = units[k0 : k0 + args.lanes]
            split = max(1, args.lanes // len(batch))  # spare lane slots: cut each path into contiguous ranges
            for f, r in batch:
                # alpha_max for this l1_ratio (upper bound: group part alone or l1 part alone)
                amax = min(bmax / (1 - r), float(np.max(np.abs(g0))) / r)
                al = np.geomspace(amax, 1e-3 * amax, 50)
                pts = np.c_[r * al, (1 - r) * al, 0 * al]
                for part in np.array_split(np.arange(50), split):
                    specs.append(dict(points=pts[part], row_weight=masks[f], n_eff=int(masks[f].sum())))
            out = ds.solve_lanes(specs)
            total_passes += out[0].grad_launches
            assert all(o.converged for o in out)
        return total_passes

    run_grid()
    t0 = time.perf_counter()
    passes = run_grid()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": f"4: SparseGroupLasso 5-fold x 10 l1_ratio x 50 alpha = 2500 fits, 1 GPU, {args.lanes} (fold, l1_ratio) units per call",
                      "fits_per_s": 2500 / dt, "seconds_per_grid": dt, "passes": passes}), flush=True)
    ds.close()


if __name__ == "__main__":
    main()
