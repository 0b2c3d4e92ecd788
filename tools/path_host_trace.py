"""Host-side account of the headline path: SLM_TRACE=2 lines (cumulative ms since the entry of slm_solve_path_lanes) and
the wall time of each call seen from Python, for a few paths on BASELINE config 2's synthetic shape."""
import os, sys, time
os.environ["SLM_TRACE"] = "2"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
import numpy as np
from sparselm_amd import _engine
n, p, K = 100_000, 5_000, 50
eng = _engine.get_engine(0)
sys.path.insert(0, ROOT)
from bench import make_coef
ds = eng.synthetic_dataset(n, p, seed=1000, coef=make_coef(p, 50, seed=0), noise_sd=10.0)
g0, _, _ = ds.gradient(None, reps=50)
amax = float(np.max(np.abs(g0)))
pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
flags = _engine.FLAG_FRESH_L
for _ in range(4):
    ds.solve_path(pts, tol=1e-8, lanes=0, flags=flags)
t = []
for _ in range(6):
    t0 = time.perf_counter()
    r = ds.solve_path(pts, tol=1e-8, lanes=0, flags=flags)
    t.append(time.perf_counter() - t0)
    sys.stderr.write(f"[py] call {1e3 * t[-1]:.3f} ms, engine wall {r.wall_ms:.3f} ms\n")
