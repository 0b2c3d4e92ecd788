#!/usr/bin/env python3
"""Per-pass trace (SLM_TRACE=3) of one soak seed's headline path.  usage: mg_trace.py seed [group]"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import soak_case
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p = 100000, 5000
seed = int(sys.argv[1])
LANES = int(os.environ.get("MG_TRACE_LANES", "0"))  # (0: the engine's choice)
coef, noise, lo, k = soak_case(seed, p)
with eng.synthetic_dataset(n, p, seed=100 + seed, coef=coef, noise_sd=noise) as ds:
    g0, _ = ds.gradient(None)
    amax = float(np.max(np.abs(g0)))
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, lo * amax, 50)]
    ds.solve_path(pts, lanes=LANES, flags=_engine.FLAG_FRESH_L)
    os.environ["SLM_TRACE"] = "3"
    t = time.perf_counter(); r = ds.solve_path(pts, lanes=LANES, flags=_engine.FLAG_FRESH_L); dt = (time.perf_counter() - t) * 1e3
    del os.environ["SLM_TRACE"]
    nnz = [int(np.count_nonzero(b)) for b in r.betas]
    print(f"seed {seed}: {dt:.2f} ms, {r.grad_launches} passes, rounds {r.mg_rounds} inner {r.mg_inner_iters}; nnz per point: {nnz}")
