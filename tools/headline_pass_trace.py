"""The headline path pass by pass (SLM_TRACE): lanes, points, working-set counters after every pass of one solve on the bench's dataset."""
import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import make_coef
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p, K = 100000, 5000, 50
dseed = int(sys.argv[1]); lanes = int(sys.argv[2])
coef = make_coef(p, 50, seed=0)
with eng.synthetic_dataset(n, p, seed=dseed, coef=coef, noise_sd=10.0) as ds:
    g0, _ = ds.gradient(None)
    amax = float(np.max(np.abs(g0)))
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
    ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L)
    os.environ["SLM_TRACE"] = "3"; os.environ["SLM_TRACE_POLL"] = "1"
    r = ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L)
    del os.environ["SLM_TRACE"]
    print("nnz per point", [int(np.count_nonzero(b)) for b in r.betas])
