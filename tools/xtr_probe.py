#!/usr/bin/env python3
"""Time the split pass kernels in isolation (slm_gradient probe with SLM_GRAD_SPLIT=1)."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd")); sys.path.insert(0, ROOT)
from sparselm_amd import _engine
from bench import make_coef
eng = _engine.get_engine(0)
n, p = 100_000, 5_000
for deep in ("0",):
    ds = eng.synthetic_dataset(n, p, seed=1000, coef=make_coef(p, 50, seed=0), noise_sd=10.0)
    os.environ.pop("SLM_GRAD_SPLIT", None)
    ds.gradient(None, reps=30)
    os.environ["SLM_GRAD_SPLIT"] = "1"
    os.environ["SLM_PROBE_LANES"] = "8"
    os.environ["SLM_GRAD_SPLIT_XTR_ONLY"] = "1"
    _, _, ms = ds.gradient(None, reps=30)
    print(f"deep={deep} xtr only: {ms:.4f} ms  -> {8*(n*p)/ms/1e6:.0f} GB/s", flush=True)
    os.environ.pop("SLM_GRAD_SPLIT_XTR_ONLY")
    _, _, ms = ds.gradient(None, reps=30)
    print(f"deep={deep} rowdot+xtr: {ms:.4f} ms", flush=True)
    os.environ.pop("SLM_GRAD_SPLIT"); os.environ.pop("SLM_PROBE_LANES")
    for lanes in ("1", "4"):
        os.environ["SLM_PROBE_LANES"] = lanes
        _, _, ms = ds.gradient(None, reps=30)
        print(f"fused lanes={lanes}: {ms:.4f} ms", flush=True)
    os.environ.pop("SLM_PROBE_LANES")
    ds.close()
