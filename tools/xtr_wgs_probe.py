#!/usr/bin/env python3
"""The pass over X of working-set solves (xtr_mfma_kernel) alone, at 2 / 1.5 / 1 / 0.75 / 0.5 workgroups per CU
(SLM_XTR_WGS_PER_CU), n = 100 000, p = 5 000, sixteen lane slots."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd")); sys.path.insert(0, ROOT)
from sparselm_amd import _engine
from bench import make_coef
eng = _engine.get_engine(0)
n, p = 100_000, 5_000
ds = eng.synthetic_dataset(n, p, seed=1000, coef=make_coef(p, 50, seed=0), noise_sd=10.0)
ds.gradient(None, reps=50)
for rep in range(3):
    for k in ("2", "1.5", "1", "0.75", "0.5"):
        os.environ["SLM_XTR_WGS_PER_CU"] = k
        _engine.reload_knobs()
        _, _, ms = ds.gradient(None, reps=40, split=True, probe_lanes=16, xtr_only=True)
        print(f"xtr only, {k} workgroups per CU: {ms:.4f} ms -> {8*n*p/ms/1e6:.0f} GB/s", flush=True)
