#!/usr/bin/env python3
"""slm_dataset_lipschitz (device power iteration, SLM_POWER_ITERS steps) against numpy on a design with a dominant
low-rank part."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p = 70000, 1200
rng = np.random.default_rng(0)
X = rng.standard_normal((n, 8)) @ rng.standard_normal((8, p)) * 2.0 + 0.3 * rng.standard_normal((n, p))
y = rng.standard_normal(n)
print("numpy lambda_max", np.linalg.norm(X, 2) ** 2 / n)
for k in (1, 2, 3, 4, 5, 6):
    os.environ["SLM_POWER_ITERS"] = str(k)
    with eng.dataset(X, y) as ds:
        print(f"iters {k}: slm_dataset_lipschitz / 1.08 = {ds.lipschitz() / 1.08:.1f}", flush=True)
