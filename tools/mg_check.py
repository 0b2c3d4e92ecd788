#!/usr/bin/env python3
"""The model Gram (csrc/mg_kernels.hpp) on the GPU: (1) its entries against numpy on small shapes (tile edges, row weights,
columns of very different scale), (2) dense-ended paths at a size the oracle finishes, with the rounds forced on (SLM_MG=2),
against the same calls without (FLAG_NO_MODEL_GRAM) and against oracle.fista, (3) at full size: three soak seeds with dense ends,
passes and wall time with and without.   usage: mg_check.py [gram|small|full ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from sparselm_amd import _engine
what = sys.argv[1:] or ["gram", "small", "full"]
eng = _engine.get_engine(0)

if "gram" in what:
    rng = np.random.default_rng(0)
    for (n, p, weighted, scales) in [(3000, 200, False, False), (1037, 77, True, False), (6500, 300, False, True), (12900, 129, True, True), (64, 16, False, False)]:
        X = rng.standard_normal((n, p))
        if scales:
            X *= 10.0 ** rng.uniform(-6, 6, p)
        y = rng.standard_normal(n)
        w = rng.uniform(0.2, 3.0, n) if weighted else None
        with eng.dataset(X, y, row_weight=w) as ds:
            t = time.perf_counter(); G = ds.model_gram(download=True); dt = (time.perf_counter() - t) * 1e3
        ld = G.shape[0]
        W = np.ones(n) if w is None else w
        ref = (X * W[:, None]).T @ X / n
        d = np.sqrt(np.diag(ref))
        rel = np.abs(G[:p, :p] - ref) / np.outer(d, d)  # entry errors relative to the columns' scales
        sym = np.max(np.abs(G - G.T))
        pad = np.max(np.abs(G[p:, :])) if ld > p else 0.0
        dn = np.diag(1.0 / d)
        spec = np.linalg.norm(dn @ (G[:p, :p] - ref) @ dn, 2)
        print(f"gram n={n} p={p} weighted={weighted} scales={scales}: max entry err {rel.max():.2e} (scaled), spectral {spec:.2e}, asym {sym:.1e}, pad {pad:.1e}, {dt:.1f} ms", flush=True)
        assert rel.max() < 2e-3 and sym == 0.0 and pad == 0.0, "model Gram off"

if "small" in what:
    import oracle
    os.environ["SLM_MG"] = "2"
    rng = np.random.default_rng(1)
    for (n, p, k, noise, lo, grouped) in [(4000, 640, 30, 100.0, 1e-3, False), (6000, 800, 60, 30.0, 1e-2, False), (5000, 600, 12, 100.0, 1e-2, True)]:
        X = rng.standard_normal((n, p))
        bt = np.zeros(p); bt[rng.choice(p, k, replace=False)] = 10 * rng.uniform(0.2, 1.0, k)
        y = X @ bt + noise * rng.standard_normal(n)
        with eng.dataset(X, y) as ds:
            if grouped:
                G = p // 10
                gid = rng.permutation(np.repeat(np.arange(G), 10)).astype(np.int32)
                ds.set_groups(gid, G)
            g0, _ = ds.gradient(None)
            if grouped:
                amax = float(np.max(np.sqrt(np.bincount(gid, weights=g0 * g0, minlength=G))))
                pts = [(0.0, a, 0.0) for a in np.geomspace(amax, lo * amax, 40)]
            else:
                amax = float(np.max(np.abs(g0)))
                pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, lo * amax, 40)]
            r = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_WORKING_SET, tol=1e-9)
            q = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_WORKING_SET | _engine.FLAG_NO_MODEL_GRAM, tol=1e-9)
            err = float(np.max(np.abs(r.betas - q.betas)) / np.max(np.abs(q.betas)))
            gidx, Gn = oracle.group_index(gid if grouped else None, p)
            worst = 0.0
            for kk in (len(pts) // 2, len(pts) - 1):
                a, b, d = pts[kk]
                bo, _ = oracle.fista(X, y, a, b, d, gidx, Gn, beta0=q.betas[kk], tol=1e-13)
                worst = max(worst, float(np.max(np.abs(r.betas[kk] - bo)) / np.max(np.abs(bo))))
            nnz = int(np.count_nonzero(r.betas[-1]))
            print(f"small n={n} p={p} grouped={grouped}: passes {r.grad_launches} (without: {q.grad_launches}), rounds {r.mg_rounds}, inner {r.mg_inner_iters}, "
                  f"rejected {r.mg_rejected}, nnz_last {nnz}, conv {r.converged}/{q.converged}, vs without {err:.1e}, vs oracle {worst:.1e}", flush=True)
            assert r.converged and err < 1e-6 and worst < 1e-6
    del os.environ["SLM_MG"]

if "full" in what:
    from bench import soak_case
    n, p = 100000, 5000
    for seed in (8, 11, 19, 53):
        coef, noise, lo, k = soak_case(seed, p)
        with eng.synthetic_dataset(n, p, seed=100 + seed, coef=coef, noise_sd=noise) as ds:
            g0, _ = ds.gradient(None)
            amax = float(np.max(np.abs(g0)))
            pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, lo * amax, 50)]
            t = time.perf_counter(); r = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_FRESH_L); dt = (time.perf_counter() - t) * 1e3
            t = time.perf_counter(); r2 = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_FRESH_L); dt2 = (time.perf_counter() - t) * 1e3
            t = time.perf_counter(); q = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_FRESH_L | _engine.FLAG_NO_MODEL_GRAM); dq = (time.perf_counter() - t) * 1e3
            ref = ds.solve_path(pts, lanes=4, flags=_engine.FLAG_NO_WORKING_SET, tol=1e-10)
            err = float(np.max(np.abs(r.betas - ref.betas)) / np.max(np.abs(ref.betas)))
            errq = float(np.max(np.abs(q.betas - ref.betas)) / np.max(np.abs(ref.betas)))
            nnz = int(np.count_nonzero(r.betas[-1]))
            print(f"full seed {seed} noise={noise} lo={lo}: {dt:.2f} ms / {r.grad_launches} passes first (build {r.mg_build_ms:.2f} ms), {dt2:.2f} ms / {r2.grad_launches} passes again, "
                  f"without {dq:.2f} ms / {q.grad_launches} passes; rounds {r.mg_rounds} inner {r.mg_inner_iters} rejected {r.mg_rejected}; nnz_last {nnz}; "
                  f"err {err:.1e} (without: {errq:.1e}) conv {r.converged}", flush=True)
print("MG CHECK done")
