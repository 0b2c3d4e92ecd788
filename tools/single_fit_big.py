"""One cold Lasso solve (one alpha) on the headline's data (100 000 x 5 000): passes and time, with the sample start and
without (SLM_NO_SAMPLE_START=1)."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
import numpy as np
from bench import make_coef
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p = 100_000, 5_000
k_inf = int(sys.argv[1]) if len(sys.argv) > 1 else 50  # informative features
ds = eng.synthetic_dataset(n, p, seed=1000, coef=make_coef(p, k_inf, seed=0), noise_sd=10.0)
g0, _, _ = ds.gradient(None, reps=20)
amax = float(np.max(np.abs(g0)))
for frac in (0.3, 0.05, 0.005):
    for env in (None, "1"):
        if env:
            os.environ["SLM_NO_SAMPLE_START"] = env
        else:
            os.environ.pop("SLM_NO_SAMPLE_START", None)
        pts = [(frac * amax, 0.0, 0.0)]
        for _ in range(3):
            r = ds.solve_path(pts, tol=1e-8)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            r = ds.solve_path(pts, tol=1e-8)
            ts.append(time.perf_counter() - t0)
        print(f"alpha = {frac} alpha_max, sample start {'off' if env else 'on '}: {r.grad_launches} passes, {1e3 * np.median(ts):.3f} ms, nnz {int(np.count_nonzero(r.betas[-1]))}, converged {r.converged}", flush=True)
