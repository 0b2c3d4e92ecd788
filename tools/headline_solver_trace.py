"""The model solver of the headline path under SLM_TRACE=2: iterations per refinement, lane 0's phases, ms per lane."""
import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import make_coef
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p, K = 100000, 5000, 50
coef = make_coef(p, 50, seed=0)
with eng.synthetic_dataset(n, p, seed=1000, coef=coef, noise_sd=10.0) as ds:
    g0, _ = ds.gradient(None)
    amax = float(np.max(np.abs(g0)))
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
    ds.solve_path(pts, lanes=0, flags=_engine.FLAG_FRESH_L)
    os.environ["SLM_TRACE"] = "2"
    r = ds.solve_path(pts, lanes=0, flags=_engine.FLAG_FRESH_L)
    del os.environ["SLM_TRACE"]
    print("passes", r.grad_launches)
    print("nnz per point", [int(np.count_nonzero(b)) for b in r.betas])
