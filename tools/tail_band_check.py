#!/usr/bin/env python3
"""The points beyond the last full band of a shared path (48, 49 of 50 on sixteen lanes): given to the last lanes
(default) or to the first (SLM_NO_TAIL_BAND=1) -- passes, misses, columns, wall time, agreement."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd")); sys.path.insert(0, ROOT)
import numpy as np
from sparselm_amd import _engine
from bench import make_coef
eng = _engine.get_engine(0)
n, p = 100_000, 5_000
for seed in (1000, 1001, 1002):
    ds = eng.synthetic_dataset(n, p, seed=seed, coef=make_coef(p, 50, seed=0), noise_sd=10.0)
    g0, _ = ds.gradient(None)
    amax = float(np.max(np.abs(g0)))
    out = {}
    for K in (50, 45, 64):
        pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
        for mode in ("first", "last"):
            if mode == "first":
                os.environ["SLM_NO_TAIL_BAND"] = "1"
            else:
                os.environ.pop("SLM_NO_TAIL_BAND", None)
            ds.solve_path(pts, lanes=16, flags=_engine.FLAG_FRESH_L)
            w = []
            for _ in range(5):
                r = ds.solve_path(pts, lanes=16, flags=_engine.FLAG_FRESH_L)
                w.append(r.wall_ms)
            out[mode] = r
            print(f"seed {seed} K={K} tail band to the {mode:5s} lanes: {np.median(w):.3f} ms, {r.grad_launches} passes, b/a/m/cols "
                  f"{r.ws_builds}/{r.ws_appends}/{r.ws_misses}/{r.ws_columns}, converged {r.converged}", flush=True)
        print("   agreement", float(np.max(np.abs(out['first'].betas - out['last'].betas)) / np.max(np.abs(out['first'].betas))))
    ds.close()
