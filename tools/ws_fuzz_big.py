#!/usr/bin/env python3
"""A few random problems just above the large-X threshold (n * ld >= 2^26: split pass and working set by default,
sixteen lanes), every penalty kind, against the plain one-lane iteration."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, bad = 0.0, 0
for case in range(n_cases):
    p = int(rng.integers(3000, 5121)); n = int(2**26 // (((p + 15) // 16) * 16) + rng.integers(1, 3000))
    kind = rng.choice(["lasso", "group", "sgl", "ridged", "wl1"])
    gsz = int(rng.integers(2, 12)); G = max(1, p // gsz)
    groups = rng.permutation(np.arange(p) % G) if kind in ("group", "sgl", "ridged") else None
    coef = np.zeros(p); nz = rng.choice(p, int(rng.integers(3, 60)), replace=False); coef[nz] = rng.standard_normal(len(nz)) * 5
    with eng.synthetic_dataset(n, p, seed=500 + case, coef=coef, noise_sd=float(rng.choice([0.5, 5.0]))) as ds:
        if groups is not None: ds.set_groups(groups, G)
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0))) if groups is None else float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G))))
        K = int(rng.integers(6, 30)); al = np.geomspace(amax, float(rng.choice([0.2, 0.03])) * amax, K)
        a = rng.uniform(0.5, 2.0, p) if kind == "wl1" else None
        b = rng.uniform(0.5, 2.0, G) if groups is not None else None
        pts = {"lasso": [(x, 0, 0) for x in al], "wl1": [(x, 0, 0) for x in al], "group": [(0, x, 0) for x in al],
               "sgl": [(0.4 * x, 0.6 * x, 0) for x in al], "ridged": [(0, x, 0.3) for x in al]}[kind]
        lanes = int(rng.integers(2, 17))
        if rng.random() < 0.5:
            t = time.perf_counter(); r1 = ds.solve_path(pts, a=a, b=b, tol=1e-10, lanes=lanes); dt = time.perf_counter() - t
            r0 = ds.solve_path(pts, a=a, b=b, tol=1e-10, lanes=1, flags=_engine.FLAG_NO_WORKING_SET)
            B1, B0, ok, passes = r1.betas, r0.betas, r1.converged and r0.converged, r1.grad_launches
        else:
            fold = rng.integers(0, 4, n)
            specs = [dict(points=pts, a=a, b=b, row_weight=(fold != f % 4).astype(float), n_eff=int(np.sum(fold != f % 4))) for f in range(lanes)]
            t = time.perf_counter(); R1 = ds.solve_lanes(specs, tol=1e-10); dt = time.perf_counter() - t
            pick = sorted(set([0, lanes // 2, lanes - 1]))
            R0 = [ds.solve_lanes([specs[f]], tol=1e-10, flags=_engine.FLAG_NO_WORKING_SET)[0] for f in pick]
            B1 = np.stack([R1[f].betas for f in pick]); B0 = np.stack([r.betas for r in R0])
            ok = all(r.converged for r in R1) and all(r.converged for r in R0); passes = R1[0].grad_launches
    err = float(np.max(np.abs(B1 - B0)) / max(np.max(np.abs(B0)), 1e-300))
    worst = max(worst, err); flag = "" if ok and err < 1e-6 else "  <-- CHECK"; bad += bool(flag)
    print(f"{case:2d} n={n} p={p} {kind:6s} gsz={gsz:2d} K={K:2d} lanes={lanes:2d}: {dt*1e3:7.1f} ms {passes:3d} passes err={err:.1e}{flag}", flush=True)
print(f"BIGFUZZ cases {n_cases} worst {worst:.2e} flagged {bad}")
