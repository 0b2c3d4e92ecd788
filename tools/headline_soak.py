#!/usr/bin/env python3
"""The headline path (n=100k, p=5k, 50 alphas, the engine's choice of lanes, working set) on many random datasets: converged, pass
count, and agreement with the plain 4-lane iteration of the same data."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import soak_case  # (the law of the datasets: shared with bench.py's `soak` leg)
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p = 100000, 5000
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
LANES = int(os.environ.get("SOAK_LANES", "0"))  # (0: the engine's choice -- eighteen for fifty points)
worst, slow, bad = 0.0, 0, 0
for seed in range(seeds):
    coef, noise, lo, k = soak_case(seed, p)
    with eng.synthetic_dataset(n, p, seed=100 + seed, coef=coef, noise_sd=noise) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, lo * amax, 50)]
        first = time.perf_counter(); ds.solve_path(pts, lanes=LANES, flags=_engine.FLAG_FRESH_L); first = (time.perf_counter() - first) * 1e3  # (one-off costs of a fresh dataset, the model Gram's build included)
        t = time.perf_counter(); r = ds.solve_path(pts, lanes=LANES, flags=_engine.FLAG_FRESH_L); dt = (time.perf_counter() - t) * 1e3
        q = ds.solve_path(pts, lanes=4, flags=_engine.FLAG_NO_WORKING_SET, tol=1e-9)
        err = float(np.max(np.abs(r.betas - q.betas)) / max(np.max(np.abs(q.betas)), 1e-300))
        nnz = int(np.count_nonzero(r.betas[-1]))
        worst = max(worst, err)
        flag = "" if (r.converged and q.converged and err < 1e-6) else "  <-- CHECK"
        bad += bool(flag); slow += r.grad_launches > 7
        print(f"seed {seed:2d} k={k:3d} noise={noise:5.1f} lo={lo:5.3f}: {dt:6.2f} ms (first solve {first:6.2f}), {r.grad_launches:2d} passes (plain: {q.grad_launches}), ws b/a/m/cols {r.ws_builds}/{r.ws_appends}/{r.ws_misses}/{r.ws_columns}, nnz_last {nnz}, err {err:.1e}{flag}", flush=True)
print(f"SOAK seeds {seeds} worst rel-inf {worst:.2e} flagged {bad} paths over 7 passes {slow}")
