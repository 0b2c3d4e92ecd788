"""Randomised cross-check of covariance passes (SLM_FLAG_COVARIANCE): the same lanes -- penalty family, group sizes, lane
counts, row sets (folds as 0/1 masks, general weights, none), per-lane 1/n, warm starts, p > n -- solved over X and from
the Grams of their row sets.  `python tools/covariance_fuzz.py [cases] [seed]`"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sparse-lm_amd"))
from sparselm_amd import _engine  # noqa: E402

eng = _engine.get_engine(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, flagged, t_x, t_c = 0.0, 0, 0.0, 0.0
for case in range(cases):
    n = int(rng.integers(300, 5000))
    p = int(rng.integers(40, 700))
    if rng.random() < 0.15:
        n = int(rng.integers(40, p))  # p > n
    gsz = int(rng.choice([1, 1, 2, 5, 10]))
    G = (p + gsz - 1) // gsz
    groups = rng.permutation(np.arange(p) % G) if gsz > 1 else None
    X = rng.standard_normal((n, p))
    if rng.random() < 0.3:
        X += rng.uniform(0.3, 2.0) * rng.standard_normal((n, 1))
    coef = np.where(rng.random(p) < 0.1, 3.0 * rng.standard_normal(p), 0.0)
    y = X @ coef + rng.uniform(0.1, 3.0) * rng.standard_normal(n)
    kind = str(rng.choice(["lasso", "group", "sgl", "ridged"])) if gsz > 1 else "lasso"
    lanes = int(rng.integers(1, 33))  # (thirty-two on working-set solves: both halves against one read of every Gram)
    K = int(rng.integers(1, 9))
    nsets = int(rng.integers(1, 4))
    sets = []
    for s in range(nsets):
        r = rng.random()
        if r < 0.5:
            f = int(rng.integers(3, 7))
            w = (rng.permutation(n) % f != 0).astype(float)
            sets.append((w, int(w.sum())))
        elif r < 0.75:
            sets.append((rng.uniform(0.2, 2.0, n), 0))
        else:
            sets.append((None, 0))
    with eng.dataset(X, y) as ds:
        if groups is not None:
            ds.set_groups(groups, G)
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        specs = []
        for l in range(lanes):
            al = np.geomspace(amax * rng.uniform(0.3, 1.0), amax * rng.uniform(0.01, 0.1), K)
            pts = {"lasso": np.c_[al, 0 * al, 0 * al], "group": np.c_[0 * al, 2 * al, 0 * al], "sgl": np.c_[0.5 * al, al, 0 * al],
                   "ridged": np.c_[0 * al, 2 * al, 0.3 + 0 * al]}[kind]
            w, ne = sets[l % nsets]
            spec = dict(points=pts, row_weight=w, n_eff=ne)
            if rng.random() < 0.2:
                spec["beta0"] = 0.1 * rng.standard_normal(p)
            specs.append(spec)
        base = _engine.FLAG_WORKING_SET if rng.random() < 0.7 else _engine.FLAG_NO_WORKING_SET
        try:
            t0 = time.perf_counter()
            ref = ds.solve_lanes(specs, tol=1e-10, max_iter=200000, flags=base)
            t_x += time.perf_counter() - t0
        except NotImplementedError:
            print(f"{case:3d} n={n:5d} p={p:4d} {kind:6s} lanes={lanes:2d}: no kernel serves this call over X; skipped", flush=True)
            continue
        for w, ne in sets:
            ds.covariance(w, ne)
        t0 = time.perf_counter()
        cov = ds.solve_lanes(specs, tol=1e-10, max_iter=200000, flags=base | _engine.FLAG_COVARIANCE)
        t_c += time.perf_counter() - t0
        err = 0.0
        for a, b in zip(ref, cov):
            scale = max(float(np.max(np.abs(a.betas))), 1e-300)
            err = max(err, float(np.max(np.abs(a.betas - b.betas))) / scale)
        ok = all(r.converged for r in ref) and all(r.converged for r in cov)
        # (p > n: the minimiser need not be unique; both runs are then judged by their objectives' agreement through the loss)
        bad = (not ok) or (err > 1e-6 and n > p)
        flagged += bad
        worst = max(worst, err if n > p else 0.0)
        print(f"{case:3d} n={n:5d} p={p:4d} {kind:6s} gsz={gsz:2d} K={K} lanes={lanes:2d} sets={nsets} ws={base == _engine.FLAG_WORKING_SET} "
              f"passes {ref[0].grad_launches:4d}/{cov[0].grad_launches:4d} conv={ok} err={err:.2e}{'  <-- CHECK' if bad else ''}", flush=True)
print(f"FUZZ cases {cases}  worst rel-inf {worst:.2e}  flagged {flagged}  ({t_x:.1f} s over X, {t_c:.1f} s from the Grams)")
sys.exit(1 if flagged else 0)
