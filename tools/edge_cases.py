"""Odd shapes through the round-3 kernels: one to three features, two rows, one group over everything, p = 128, paths
of one point on sixteen lanes -- on chip, through the splitting, and from Grams -- each against the general path."""
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sparse-lm_amd"))
from sparselm_amd import _engine  # noqa: E402

warnings.simplefilter("ignore")
eng = _engine.get_engine(0)
rng = np.random.default_rng(0)
worst = 0.0
for n, p, gsz in ((2, 1, 1), (3, 2, 2), (5, 3, 1), (2, 7, 7), (40, 128, 8), (40, 128, 128), (300, 17, 1), (1000, 128, 4), (9, 9, 3), (64, 65, 5)):
    X = rng.standard_normal((n, p))
    y = X @ rng.standard_normal(p) + 0.1 * rng.standard_normal(n)
    G = (p + gsz - 1) // gsz
    groups = np.arange(p) // gsz
    with eng.dataset(X, y) as ds:
        ds.set_groups(groups if gsz > 1 else None, G if gsz > 1 else None)
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0))) or 1.0
        for kind, pts in (("lasso", [(0.3 * amax, 0, 0), (0.03 * amax, 0, 0)]), ("group", [(0, 0.5 * amax, 0), (0, 0.05 * amax, 0)]),
                          ("sgl+ridge", [(0.1 * amax, 0.2 * amax, 0.1)])):
            specs = [dict(points=np.array(pts) * s) for s in np.linspace(0.5, 1.5, 16)]
            ref = ds.solve_lanes(specs[:4], tol=1e-10, max_iter=400000)
            chip = ds.solve_lanes(specs, tol=1e-10, max_iter=400000, flags=_engine.FLAG_ON_CHIP)
            f = lambda b, q: 0.5 * np.mean((X @ b - y) ** 2) + q[0] * np.abs(b).sum() + q[1] * sum(np.linalg.norm(b[groups == g]) for g in range(G)) + 0.5 * q[2] * sum(b[groups == g] @ b[groups == g] for g in range(G))  # noqa: E731
            for l in range(4):
                for k in range(len(pts)):
                    q = np.array(pts[k]) * np.linspace(0.5, 1.5, 16)[l]
                    fa, fb = f(ref[l].betas[k], q), f(chip[l].betas[k], q)
                    worst = max(worst, abs(fa - fb) / max(abs(fa), 1e-300))
                    assert chip[l].converged and abs(fa - fb) <= 1e-8 * max(abs(fa), 1e-12), (n, p, gsz, kind, l, k, fa, fb)
        # the splitting for the standardised sparse-group penalty, where the kernel takes the problem
        try:
            coef, gn, rec = ds.solve_standardized_sgl(0.05 * amax * np.ones(p), 0.05 * amax * np.ones(G), tol=1e-10, max_sweeps=5000, want_group_norms=True)
            note = f"splitting on chip: {rec['n_iter']} sweeps, status {rec['status']}"
            assert np.all(np.isfinite(coef))
        except NotImplementedError as exc:
            note = "splitting: " + str(exc)[:60]
        # covariance passes (the split pass has to exist for the shape)
        try:
            ds.covariance(None, 0)
            a = ds.solve_lanes(specs[:3], tol=1e-10, max_iter=400000, flags=_engine.FLAG_WORKING_SET)
            b = ds.solve_lanes(specs[:3], tol=1e-10, max_iter=400000, flags=_engine.FLAG_WORKING_SET | _engine.FLAG_COVARIANCE)
            d = max(float(np.max(np.abs(u.betas - v.betas))) / max(float(np.max(np.abs(u.betas))), 1e-300) for u, v in zip(a, b))
            note += f"; Grams vs X {d:.1e}"
            assert d < 1e-6 or n < p
        except NotImplementedError as exc:
            note += "; Grams: " + str(exc)[:50]
    print(f"n={n:5d} p={p:4d} group size {gsz:3d}: ok ({note})", flush=True)
print(f"edge cases done, worst objective gap on chip vs general {worst:.2e}")
