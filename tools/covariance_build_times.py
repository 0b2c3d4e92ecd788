import sys, time
sys.path.insert(0, "/root/repo/sparse-lm_amd")
import numpy as np
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p = 100_000, 5_000
ds = eng.synthetic_dataset(n, p, seed=3, coef=np.ones(p) * 0.01, noise_sd=1.0)
rng = np.random.default_rng(0)
folds = rng.permutation(n) % 5
for f in range(5):
    m = (folds != f).astype(float)
    t0 = time.perf_counter(); ds.covariance(m, int(m.sum())); print(f"fold {f}: {time.perf_counter()-t0:.3f} s", flush=True)
t0 = time.perf_counter(); ds.covariance(None, 0); print(f"all rows: {time.perf_counter()-t0:.3f} s", flush=True)
ds.close()
