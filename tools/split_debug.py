#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
eng = _engine.get_engine(0)
for n, p in ((700, 5000), (513, 5120), (700, 4990), (256, 5000), (2000, 5000), (700, 4100)):
    rng = np.random.default_rng(n * 31 + p)
    X = rng.standard_normal((n, p)); y = rng.standard_normal(n); z = rng.standard_normal(p)
    g0 = X.T @ (X @ z - y) / n
    with eng.dataset(X, y) as ds:
        os.environ["SLM_GRAD_SPLIT"] = "1"
        g, loss = ds.gradient(z)
        os.environ.pop("SLM_GRAD_SPLIT")
        gf, _ = ds.gradient(z)
    bad = np.where(np.abs(g - g0) > 1e-9 * np.max(np.abs(g0)))[0]
    print(n, p, "split err", np.max(np.abs(g - g0)) / np.max(np.abs(g0)), "fused err", np.max(np.abs(gf - g0)) / np.max(np.abs(g0)),
          "loss", loss, 0.5 * np.sum((X @ z - y) ** 2) / n, "nbad", len(bad), bad[:12], bad[-4:] if len(bad) else "")
