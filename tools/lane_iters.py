import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "sparse-lm_amd"))
import numpy as np
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p = 100000, 5000
rng = np.random.default_rng(0)
coef = np.zeros(p); idx = rng.choice(p, 50, replace=False); coef[idx] = 100 * rng.uniform(size=50)
ds = eng.synthetic_dataset(n, p, seed=1000, coef=coef, noise_sd=10.0)
g0, _ = ds.gradient(None)
amax = np.max(np.abs(g0)); alphas = np.geomspace(amax, 1e-3 * amax, 50)
pts = [(a, 0, 0) for a in alphas]
for lanes in (1, 2, 3, 4):
    r = ds.solve_path(pts, lanes=lanes)
    print("lanes", lanes, "passes", r.grad_launches)
    print("  n_iter", list(r.n_iter))
