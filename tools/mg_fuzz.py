#!/usr/bin/env python3
"""Randomised cross-check of the model-Gram rounds (csrc/mg_kernels.hpp), forced on at sizes the plain iteration finishes
(SLM_MG=2): penalty kinds, group sizes, 9-32 lanes (shared paths and independent lanes with fold masks), designs that are
correlated / have duplicated columns / more columns than rows, dataset row weights, paths that end dense.  Every call is
compared with the same call without the rounds (FLAG_NO_MODEL_GRAM) and with one plain lane: 1e-6 rel-inf, or -- where the
minimiser is not unique -- the same objective.   usage: mg_fuzz.py [cases] [seed]"""
import os, sys, time
os.environ["SLM_MG"] = "2"
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
eng = _engine.get_engine(0)
WS, PLAIN, NOMG = _engine.FLAG_WORKING_SET, _engine.FLAG_NO_WORKING_SET, _engine.FLAG_NO_MODEL_GRAM
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst, bad, with_rounds, rejected = 0.0, 0, 0, 0
t0 = time.perf_counter()
for case in range(n_cases):
    n = int(rng.integers(80, 2500)); p = int(rng.integers(20, 1100))
    kind = rng.choice(["lasso", "group", "sgl", "ridged", "wl1"])
    gsz = int(rng.integers(1, 12))
    G = max(1, p // gsz)
    groups = rng.permutation(np.arange(p) % G) if kind in ("group", "sgl", "ridged") else None
    X = rng.standard_normal((n, p))
    design = rng.choice(["iid", "iid", "mixed", "ar1", "dup", "scaled"])
    if design == "mixed":
        X = X @ (np.eye(p) + 0.3 * rng.standard_normal((p, p)) / np.sqrt(p))
    elif design == "ar1":
        rho = 0.9
        for j in range(1, p):
            X[:, j] = rho * X[:, j - 1] + np.sqrt(1 - rho * rho) * X[:, j]
    elif design == "dup" and p > 4:
        X[:, p // 2] = X[:, 1]
        X[:, p - 1] = -X[:, 2]
    elif design == "scaled":
        X *= 10.0 ** rng.uniform(-3, 3, p)
    k = min(p, int(rng.integers(1, 40)))
    beta = np.zeros(p); beta[rng.choice(p, k, replace=False)] = rng.standard_normal(k) * 3
    y = X @ (beta / np.maximum(np.sqrt(np.mean(X * X, axis=0)), 1e-300)) + rng.standard_normal(n) * rng.choice([0.01, 1.0, 30.0])
    w_ds = rng.uniform(0.2, 2.0, n) if rng.random() < 0.25 else None
    W = np.ones(n) if w_ds is None else w_ds
    c = X.T @ (W * y) / n
    amax = np.max(np.abs(c)) if groups is None else np.max(np.sqrt(np.bincount(groups, weights=c * c, minlength=G)))
    K = int(rng.integers(2, 26)); lo = rng.choice([0.1, 0.01, 1e-3])
    alphas = np.geomspace(amax, lo * amax, K)
    a = rng.uniform(0.5, 2.0, p) if kind == "wl1" else None
    b = rng.uniform(0.5, 2.0, G) if groups is not None else None
    if kind in ("lasso", "wl1"): pts = [(al, 0, 0) for al in alphas]
    elif kind == "group": pts = [(0, al, 0) for al in alphas]
    elif kind == "sgl": pts = [(0.4 * al, 0.6 * al, 0) for al in alphas]
    else: pts = [(0, al, 0.3) for al in alphas]
    lanes = int(rng.integers(9, 33))  # (above sixteen: two halves on one read of X)
    tol = 1e-10
    with eng.dataset(X, y, row_weight=w_ds) as ds:
        if groups is not None: ds.set_groups(groups, G)
        shared = rng.random() < 0.6 or w_ds is not None
        if shared:
            r1 = ds.solve_path(pts, a=a, b=b, tol=tol, max_iter=300000, lanes=lanes, flags=WS)
            r2 = ds.solve_path(pts, a=a, b=b, tol=tol, max_iter=300000, lanes=lanes, flags=WS | NOMG)
            r0 = ds.solve_path(pts, a=a, b=b, tol=tol, max_iter=300000, lanes=1, flags=PLAIN)
            B1, B2, B0, ok = r1.betas, r2.betas, r0.betas, r1.converged and r2.converged and r0.converged
            st, st2 = r1, r2
        else:  # independent lanes with fold masks: a model Gram per row set
            nl = lanes
            nf = int(rng.integers(2, 5))
            fold = rng.integers(0, nf, n)
            masks = [(fold != f).astype(float) for f in range(nf)]
            specs = [dict(points=pts[: max(2, K - (l % 3))], a=a, b=b, row_weight=masks[l % nf], n_eff=int(masks[l % nf].sum())) for l in range(nl)]
            R1 = ds.solve_lanes(specs, tol=tol, max_iter=300000, flags=WS)
            R2 = ds.solve_lanes(specs, tol=tol, max_iter=300000, flags=WS | NOMG)
            R0 = [ds.solve_lanes([s], tol=tol, max_iter=300000, flags=PLAIN)[0] for s in specs[:4]]
            B1 = np.concatenate([r.betas for r in R1]); B2 = np.concatenate([r.betas for r in R2])
            B0 = np.concatenate([r.betas for r in R0]); ok = all(r.converged for r in R1 + R2 + R0)
            st, st2 = R1[0], R2[0]
    with_rounds += st.mg_rounds > 0
    rejected += st.mg_rejected
    scale = max(np.max(np.abs(B2)), 1e-300)
    err = float(np.max(np.abs(B1 - B2)) / scale)
    err0 = float(np.max(np.abs(B1[: len(B0)] - B0)) / max(np.max(np.abs(B0)), 1e-300))
    worst = max(worst, err, err0)
    flag = "" if (ok and err < 1e-6 and err0 < 1e-6) else "  <-- CHECK"
    if flag and shared:
        def obj(Bm):
            out = []
            for (a1, b1, d1), bt in zip(pts, Bm):
                r = X @ bt - y
                f = 0.5 * np.sum(W * r * r) / n
                av = np.ones(p) if a is None else a
                f += a1 * np.sum(av * np.abs(bt))
                if groups is not None:
                    gn = np.sqrt(np.bincount(groups, weights=bt * bt, minlength=G))
                    f += b1 * np.sum(b * gn) + 0.5 * d1 * np.sum(gn * gn)
                out.append(f)
            return np.array(out)
        o1, o0 = obj(B1), obj(B0)
        if ok and np.max((o1 - o0) / o0) < 1e-10:
            flag = "  (flat objective: equal to %.1e)" % np.max(np.abs(o1 - o0) / o0)
        else:
            bad += 1
        print("     objective(rounds)-objective(plain), relative: max %.2e min %.2e" % (np.max((o1 - o0) / o0), np.min((o1 - o0) / o0)))
    elif flag:
        bad += 1
    print(f"{case:3d} n={n:5d} p={p:4d} {kind:6s} {design:6s} gsz={gsz:2d} K={K:2d} lanes={lanes:2d} {'path ' if shared else 'folds'} rw={w_ds is not None} "
          f"rounds={st.mg_rounds} inner={st.mg_inner_iters} rej={st.mg_rejected} passes={st.grad_launches} (without: {st2.grad_launches}) conv={ok} err={err:.2e} vs plain={err0:.2e}{flag}", flush=True)
print(f"MG FUZZ cases {n_cases}  with rounds {with_rounds}  rejected proposals {rejected}  worst rel-inf {worst:.2e}  flagged {bad}  ({time.perf_counter()-t0:.1f} s)")
