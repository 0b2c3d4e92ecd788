#!/usr/bin/env python3
"""Pass-by-pass trace (SLM_TRACE=3, every pass polled) of the headline path on given draws of the headline's law: which lane
stands at which point after each pass, the working set's counters -- where does a fourth pass come from?
usage: draw_trace.py [data seeds ...]"""
import os, sys
os.environ["SLM_TRACE"] = "3"
os.environ["SLM_TRACE_POLL"] = "1"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
import numpy as np
from bench import make_coef
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p, K = 100000, 5000, 50
coef = make_coef(p, 50, seed=0)
for dseed in [int(s) for s in sys.argv[1:]] or [7, 1000]:
    with eng.synthetic_dataset(n, p, seed=dseed, coef=coef, noise_sd=10.0) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        alphas = np.geomspace(amax, 1e-3 * amax, K)
        pts = [(a, 0.0, 0.0) for a in alphas]
        sys.stderr.write(f"=== data seed {dseed}\n")
        r = ds.solve_path(pts, lanes=0, flags=_engine.FLAG_FRESH_L)
        nnz = [int(np.count_nonzero(b)) for b in r.betas]
        sys.stderr.write(f"passes {r.grad_launches} light {r.light_passes} (columns {r.light_columns}) builds {r.ws_builds} appends {r.ws_appends} misses {r.ws_misses} columns {r.ws_columns}\nnnz per point {nnz}\n")
        # which features enter where, and how they rank in the gradient at zero
        order = np.argsort(-np.abs(g0))
        rank = np.empty(p, int); rank[order] = np.arange(p)
        first = {}
        for k, b in enumerate(r.betas):
            for j in np.flatnonzero(b):
                first.setdefault(int(j), k)
        late = sorted((k, j, int(rank[j]), float(coef[j])) for j, k in first.items())
        sys.stderr.write("entering (point, feature, rank of |g0|, true coef): " + " ".join(f"{k}:{j}/r{rk}/{c:.1f}" for k, j, rk, c in late) + "\n")
