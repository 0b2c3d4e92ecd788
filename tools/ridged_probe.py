import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "sparse-lm_amd"))
from sparselm_amd import _engine
eng = _engine.get_engine(0)
rng = np.random.default_rng(5)
n, p, gsz = 15016, 4580, 6
G = p // gsz
groups = rng.permutation(np.arange(p) % G)
coef = np.zeros(p); nz = rng.choice(p, 30, replace=False); coef[nz] = rng.standard_normal(30) * 5
with eng.synthetic_dataset(n, p, seed=512, coef=coef, noise_sd=5.0) as ds:
    ds.set_groups(groups, G)
    g0, _ = ds.gradient(None)
    amax = float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G))))
    al = np.geomspace(amax, 0.03 * amax, 10)
    b = rng.uniform(0.5, 2.0, G)
    for d in (0.0, 0.3):
        pts = [(0, x, d) for x in al]
        for lanes in (1, 10):
            r = ds.solve_path(pts, b=b, tol=1e-10, lanes=lanes)
            print(f"ridge={d} shared path lanes={lanes}: {r.grad_launches} passes, ws b/a/r/m/cols {r.ws_builds}/{r.ws_appends}/{r.ws_refined}/{r.ws_misses}/{r.ws_columns}, n_iter {r.n_iter.tolist()}")
        fold = rng.integers(0, 4, n)
        specs = [dict(points=pts, b=b, row_weight=(fold != f % 4).astype(float), n_eff=int(np.sum(fold != f % 4))) for f in range(10)]
        R = ds.solve_lanes(specs, tol=1e-10)
        print(f"ridge={d} 10 fold lanes: {R[0].grad_launches} passes, ws b/a/r/m/cols {R[0].ws_builds}/{R[0].ws_appends}/{R[0].ws_refined}/{R[0].ws_misses}/{R[0].ws_columns}, n_iter lane0 {R[0].n_iter.tolist()} lane3 {R[3].n_iter.tolist()}")
