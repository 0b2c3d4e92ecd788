#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from sparselm_amd import _engine
from test_working_set_gpu import problem, alpha_path
eng = _engine.get_engine(0)
n, p, lanes = 2500, 333, 3
X, y = problem(n, p, min(12, p // 2), seed=n + p)
alphas = alpha_path(X, y)
pts = [(a, 0, 0) for a in alphas]
for split in ("1", "0"):
    os.environ["SLM_SPLIT"] = split
    with eng.dataset(X, y) as ds:
        for ln in (1, 3, 4, 8):
            r = ds.solve_path(pts, tol=1e-11, lanes=ln, flags=_engine.FLAG_WORKING_SET)
            print("split", split, "lanes", ln, "passes", r.grad_launches, "n_iter", r.n_iter, "mode", r.mode, "ws", r.ws_builds, r.ws_appends, r.ws_refined, r.ws_misses, r.ws_columns, flush=True)
