import os, sys, time
sys.path[:0] = ['/root/repo', '/root/repo/sparse-lm_amd']
import numpy as np
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p = 100000, 5000
rng = np.random.default_rng(0)
coef = np.zeros(p); coef[rng.choice(p, 50, replace=False)] = 10 * rng.standard_normal(50)
ds = eng.synthetic_dataset(n, p, seed=1000, coef=coef, noise_sd=10.0)
g0, _ = ds.gradient(None)
amax = float(np.max(np.abs(g0)))
alphas = np.geomspace(amax, 1e-3 * amax, 50)
pts = np.c_[alphas, 0 * alphas, 0 * alphas]
flags = _engine.FLAG_FRESH_L
for _ in range(4):
    r = ds.solve_path(pts, tol=1e-8, lanes=16, flags=flags)
ts = []
for _ in range(20):
    t0 = time.perf_counter(); r = ds.solve_path(pts, tol=1e-8, lanes=16, flags=flags); ts.append(time.perf_counter() - t0)
print("per call ms: median %.3f min %.3f" % (1e3 * np.median(ts), 1e3 * min(ts)), "passes", r.grad_launches, file=sys.stderr)
os.environ["SLM_TRACE"] = "2"
for _ in range(3):
    t0 = time.perf_counter(); r = ds.solve_path(pts, tol=1e-8, lanes=16, flags=flags); dt = time.perf_counter() - t0
    print("python-side wall %.3f ms" % (1e3 * dt), file=sys.stderr)
import cProfile, pstats
os.environ["SLM_TRACE"] = "0"
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    r = ds.solve_path(pts, tol=1e-8, lanes=16, flags=flags)
pr.disable()
pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(12)
