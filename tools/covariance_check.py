"""Covariance passes against passes over X: the same sixteen-lane calls (CV folds as row masks, SparseGroupLasso paths)
with and without SLM_FLAG_COVARIANCE.  `python tools/covariance_check.py [n p]`"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sparse-lm_amd"))
from sparselm_amd import _engine  # noqa: E402

n, p = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20000, 800)
rng = np.random.default_rng(0)
G = p // 10
groups = rng.permutation(np.repeat(np.arange(G), 10))
coef = np.zeros(p)
for g in rng.choice(G, 8, replace=False):
    coef[groups == g] = 10.0 * rng.uniform(size=10)
X = rng.standard_normal((n, p))
y = X @ coef + 5.0 * rng.standard_normal(n)
folds = rng.permutation(n) % 5
masks = [(folds != f).astype(float) for f in range(5)]
eng = _engine.get_engine(0)
with eng.dataset(X, y) as ds:
    ds.set_groups(groups, G)
    g0, _ = ds.gradient(None)
    amax = float(np.max(np.abs(g0)))
    pts = np.c_[0.5 * np.geomspace(amax, 1e-2 * amax, 12), 0.5 * np.geomspace(amax, 1e-2 * amax, 12), np.zeros(12)]
    specs = [dict(points=pts * (1.0 + 0.1 * (l // 5)), row_weight=masks[l % 5], n_eff=int(masks[l % 5].sum())) for l in range(16)]
    base = _engine.FLAG_WORKING_SET
    ref = ds.solve_lanes(specs, tol=1e-10, flags=base)
    t0 = time.perf_counter()
    ref = ds.solve_lanes(specs, tol=1e-10, flags=base)
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    for m in masks:
        ds.covariance(m, int(m.sum()))
    t_build = time.perf_counter() - t0
    assert ds.covariance_count() == 5
    t0 = time.perf_counter()
    ds.covariance(masks[2], int(masks[2].sum()))  # found again: nothing built
    t_again = time.perf_counter() - t0
    cov = ds.solve_lanes(specs, tol=1e-10, flags=base | _engine.FLAG_COVARIANCE)
    t0 = time.perf_counter()
    cov = ds.solve_lanes(specs, tol=1e-10, flags=base | _engine.FLAG_COVARIANCE)
    t_cov = time.perf_counter() - t0
    worst = 0.0
    for a, b in zip(ref, cov):
        assert a.converged and b.converged
        worst = max(worst, float(np.max(np.abs(a.betas - b.betas)) / np.max(np.abs(a.betas))))
    print(f"n={n} p={p}: sixteen lanes x 12 points, passes {ref[0].grad_launches} / {cov[0].grad_launches}; {1e3 * t_ref:.2f} ms over X, "
          f"{1e3 * t_cov:.2f} ms from the Grams (built in {1e3 * t_build:.1f} ms, found again in {1e3 * t_again:.2f} ms); "
          f"worst difference {worst:.2e}")
    # general weights (not a mask) and the dataset's own rows
    w = rng.uniform(0.5, 1.5, n)
    ds.covariance(w, 0)
    ds.covariance(None, 0)
    for rw in (w, None):
        a = ds.solve_lanes([dict(points=pts, row_weight=rw)] * 2 + [dict(points=pts * 1.3, row_weight=rw)], tol=1e-10, flags=base)
        b = ds.solve_lanes([dict(points=pts, row_weight=rw)] * 2 + [dict(points=pts * 1.3, row_weight=rw)], tol=1e-10,
                           flags=base | _engine.FLAG_COVARIANCE)
        d = max(float(np.max(np.abs(u.betas - v.betas)) / np.max(np.abs(u.betas))) for u, v in zip(a, b))
        print(f"  {'general row weights' if rw is not None else 'all rows'}: worst difference {d:.2e}")
        worst = max(worst, d)
    sys.exit(0 if worst < 1e-7 else 1)
