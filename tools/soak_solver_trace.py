"""The model solver's own account (SLM_TRACE=2 clock marks) on the dense-ended cases of the soak law: iterations, direct steps, ms per lane."""
import os, sys, time
import numpy as np
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import soak_case
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p = 100000, 5000
seed = int(sys.argv[1])
coef, noise, lo, k = soak_case(seed, p)
print("case", k, noise, lo)
with eng.synthetic_dataset(n, p, seed=100 + seed, coef=coef, noise_sd=noise) as ds:
    g0, _ = ds.gradient(None)
    amax = float(np.max(np.abs(g0)))
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, lo * amax, 50)]
    ds.solve_path(pts, lanes=0, flags=_engine.FLAG_FRESH_L)
    os.environ["SLM_TRACE"] = "2"
    r = ds.solve_path(pts, lanes=0, flags=_engine.FLAG_FRESH_L)
