#!/usr/bin/env python3
"""Randomised cross-check of the on-chip solver (csrc/small_kernels.hpp, SLM_FLAG_ON_CHIP) against the general path
on problems of the reference's own sizes: penalty family, group sizes, weighted l1, ridge, fold masks, paths, p > n,
correlated columns, warm starts.  Minimisers are compared coefficient by coefficient and, where they are not unique
(p > n, duplicated columns), by objective value.  usage: on_chip_fuzz.py [cases] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, bad, fell_back = 0.0, 0, 0
t_chip = t_gen = 0.0
for case in range(n_cases):
    p = int(rng.integers(2, 129)); n = int(rng.integers(5, max(6, min(1000, 131072 // (16 * ((p + 15) // 16))))))
    kind = rng.choice(["lasso", "group", "sgl", "ridged", "wl1", "ridge_l1"])
    gsz = int(rng.integers(1, 12))
    G = max(1, p // gsz)
    groups = rng.permutation(np.arange(p) % G) if kind in ("group", "sgl", "ridged") else None
    X = rng.standard_normal((n, p))
    if rng.random() < 0.3:  # correlated columns
        X = X @ (np.eye(p) + 0.5 * rng.standard_normal((p, p)) / np.sqrt(p))
    beta = np.zeros(p); nz = rng.choice(p, min(p, int(rng.integers(1, 12))), replace=False); beta[nz] = rng.standard_normal(len(nz)) * 3
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
    elif kind == "ridged": pts = [(0, al, 0.3) for al in alphas]
    else: pts = [(al, 0, 0.5) for al in alphas]
    tol = 1e-11
    nl = int(rng.integers(1, 17))
    fold = rng.integers(0, max(2, nl), n)
    warm = rng.standard_normal(p) if rng.random() < 0.3 else None
    with eng.dataset(X, y) as ds:
        if groups is not None: ds.set_groups(groups, G)
        specs = [dict(points=pts, a=a, b=b, beta0=warm, row_weight=(fold != f).astype(float), n_eff=int(np.sum(fold != f))) for f in range(nl)]
        if rng.random() < 0.3:
            specs = [dict(points=pts, a=a, b=b, beta0=warm)]
        assert ds.max_lanes(_engine.FLAG_ON_CHIP) == _engine.MAX_CELLS
        t0 = time.perf_counter(); R1 = ds.solve_lanes(specs, tol=tol, max_iter=100000, flags=_engine.FLAG_ON_CHIP); t_chip += time.perf_counter() - t0
        t0 = time.perf_counter(); R0 = [ds.solve_lanes([s], tol=tol, max_iter=300000)[0] for s in specs]; t_gen += time.perf_counter() - t0
    on_chip = all(np.all(r.mode == 2) for r in R1)
    fell_back += not on_chip
    B1 = np.stack([r.betas for r in R1]); B0 = np.stack([r.betas for r in R0]); ok = all(r.converged for r in R1 + R0)
    scale = max(np.max(np.abs(B0)), 1e-300)
    err = float(np.max(np.abs(B1 - B0)) / scale)
    flag = "" if (ok and err < 1e-6) else "  <-- CHECK"
    if flag:
        def obj(Bm, rw, ne):
            out = []
            for (a1, b1, d1), bt in zip(pts, Bm):
                r = X @ bt - y
                f = 0.5 * np.sum(rw * r * r) / ne
                av = np.ones(p) if a is None else a
                f += a1 * np.sum(av * np.abs(bt))
                gn = np.abs(bt) if groups is None else np.sqrt(np.bincount(groups, weights=bt * bt, minlength=G))
                bv = np.ones(len(gn)) if b is None else b
                f += b1 * np.sum(bv * gn) + 0.5 * d1 * np.sum(gn * gn)
                out.append(f)
            return np.array(out)
        o1 = np.concatenate([obj(B1[i], specs[i].get("row_weight", np.ones(n)), specs[i].get("n_eff", n)) for i in range(len(specs))])
        o0 = np.concatenate([obj(B0[i], specs[i].get("row_weight", np.ones(n)), specs[i].get("n_eff", n)) for i in range(len(specs))])
        rel = (o1 - o0) / np.maximum(np.abs(o0), 1e-300)
        if ok and np.max(rel) < 1e-10:
            flag = "  (flat objective: on-chip - general = %.1e .. %.1e relative)" % (np.min(rel), np.max(rel))
        else:
            bad += 1
            flag += "  objective on-chip - general, relative: max %.2e min %.2e" % (np.max(rel), np.min(rel))
    else:
        worst = max(worst, err)
    sweeps = max(int(np.max(r.n_iter)) for r in R1)
    print(f"{case:3d} n={n:4d} p={p:3d} {kind:8s} gsz={gsz:2d} K={K:2d} lanes={len(specs):2d} on_chip={on_chip} sweeps<={sweeps} conv={ok} err={err:.2e}{flag}", flush=True)
print(f"FUZZ cases {n_cases}  worst rel-inf {worst:.2e}  flagged {bad}  handed to the general path {fell_back}  "
      f"(on-chip calls {1e3 * t_chip:.0f} ms, the same cells lane by lane on the general path {1e3 * t_gen:.0f} ms)")
sys.exit(1 if bad else 0)
