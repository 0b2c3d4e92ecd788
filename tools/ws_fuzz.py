#!/usr/bin/env python3
"""Randomised cross-check: working set + split pass against the plain iteration on many small problems
(penalty family, group sizes, lane counts, row masks, p > n)."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
eng = _engine.get_engine(0)
WS, PLAIN = _engine.FLAG_WORKING_SET, _engine.FLAG_NO_WORKING_SET
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, bad = 0.0, 0
t0 = time.perf_counter()
for case in range(n_cases):
    n = int(rng.integers(30, 1500)); p = int(rng.integers(5, 900))
    kind = rng.choice(["lasso", "group", "sgl", "ridged", "wl1"])
    gsz = int(rng.integers(1, 12))
    G = max(1, p // gsz)
    groups = rng.permutation(np.arange(p) % G) if kind in ("group", "sgl", "ridged") else None
    X = rng.standard_normal((n, p))
    if rng.random() < 0.3:  # correlated columns
        X = X @ (np.eye(p) + 0.3 * rng.standard_normal((p, p)) / np.sqrt(p))
    beta = np.zeros(p); nz = rng.choice(p, min(p, int(rng.integers(1, 15))), replace=False); beta[nz] = rng.standard_normal(len(nz)) * 3
    y = X @ beta + rng.standard_normal(n) * rng.choice([0.01, 1.0])
    c = X.T @ y / n
    amax = np.max(np.abs(c)) if groups is None else np.max(np.sqrt(np.bincount(groups, weights=c * c, minlength=G)))
    K = int(rng.integers(1, 14)); lo = rng.choice([0.3, 0.05, 0.01])
    alphas = np.geomspace(amax, lo * amax, K) if K > 1 else np.array([0.2 * amax])
    a = rng.uniform(0.5, 2.0, p) if kind == "wl1" else None
    b = rng.uniform(0.5, 2.0, G) if groups is not None else None
    if kind in ("lasso", "wl1"): pts = [(al, 0, 0) for al in alphas]
    elif kind == "group": pts = [(0, al, 0) for al in alphas]
    elif kind == "sgl": pts = [(0.4 * al, 0.6 * al, 0) for al in alphas]
    else: pts = [(0, al, 0.3) for al in alphas]
    lanes = int(rng.integers(1, 33))
    tol = 1e-10
    with eng.dataset(X, y) as ds:
        if groups is not None: ds.set_groups(groups, G)
        if rng.random() < 0.5 or K == 1:
            r1 = ds.solve_path(pts, a=a, b=b, tol=tol, max_iter=300000, lanes=lanes, flags=WS)
            r0 = ds.solve_path(pts, a=a, b=b, tol=tol, max_iter=300000, lanes=1, flags=PLAIN)
            B1, B0, ok = r1.betas, r0.betas, r1.converged and r0.converged
        else:  # independent lanes with fold masks
            nl = min(lanes, 12)
            fold = rng.integers(0, max(2, nl), n)
            specs = [dict(points=pts, a=a, b=b, row_weight=(fold != f).astype(float), n_eff=int(np.sum(fold != f))) for f in range(nl)]
            R1 = ds.solve_lanes(specs, tol=tol, max_iter=300000, flags=WS)
            R0 = [ds.solve_lanes([s], tol=tol, max_iter=300000, flags=PLAIN)[0] for s in specs]
            B1 = np.stack([r.betas for r in R1]); B0 = np.stack([r.betas for r in R0]); ok = all(r.converged for r in R1 + R0)
    scale = max(np.max(np.abs(B0)), 1e-300)
    err = float(np.max(np.abs(B1 - B0)) / scale)
    worst = max(worst, err)
    flag = "" if (ok and err < 1e-6) else "  <-- CHECK"
    if flag:
        # which of the two is closer to optimal?  (objective difference, relative)
        def obj(Bm, rw=None, ne=n):
            out = []
            for (a1, b1, d1), bt in zip(pts, Bm):
                r = X @ bt - y
                f = 0.5 * np.sum((r * r) if rw is None else rw * r * r) / ne
                av = np.ones(p) if a is None else a
                f += a1 * np.sum(av * np.abs(bt))
                if groups is not None:
                    gn = np.sqrt(np.bincount(groups, weights=bt * bt, minlength=G))
                    f += b1 * np.sum(b * gn) + 0.5 * d1 * np.sum(gn * gn)
                out.append(f)
            return np.array(out)
        if B1.ndim == 2:
            o1, o0 = obj(B1), obj(B0)
        else:
            o1 = np.concatenate([obj(B1[i], specs[i]["row_weight"], specs[i]["n_eff"]) for i in range(len(specs))])
            o0 = np.concatenate([obj(B0[i], specs[i]["row_weight"], specs[i]["n_eff"]) for i in range(len(specs))])
        if ok and np.max((o1 - o0) / o0) < 1e-10:
            flag = "  (flat objective: equal to %.1e)" % np.max(np.abs(o1 - o0) / o0)
        else:
            bad += 1
        print("     objective(ws)-objective(plain), relative: max %.2e min %.2e" % (np.max((o1 - o0) / o0), np.min((o1 - o0) / o0)))
    print(f"{case:3d} n={n:5d} p={p:4d} {kind:6s} gsz={gsz:2d} K={K:2d} lanes={lanes:2d} conv={ok} err={err:.2e}{flag}", flush=True)
print(f"FUZZ cases {n_cases}  worst rel-inf {worst:.2e}  flagged {bad}  ({time.perf_counter()-t0:.1f} s)")
sys.exit(1 if bad else 0)
