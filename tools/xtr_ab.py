#!/usr/bin/env python3
"""Mean duration of the accumulate-only pass alone (slm_gradient with SLM_GRAD_SPLIT=1, SLM_GRAD_SPLIT_XTR_ONLY=1)
for the library SLM_HIP_LIBRARY points at (default: the in-tree build): same-box A/B of two builds."""
import os, sys
os.environ["SLM_GRAD_SPLIT"] = "1"
os.environ["SLM_GRAD_SPLIT_XTR_ONLY"] = "1"
os.environ.setdefault("SLM_PROBE_LANES", "16")
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 5000
ds = eng.synthetic_dataset(n, p, seed=7, coef=np.zeros(p), noise_sd=1.0)
out = []
for _ in range(4):
    g, loss, ms = ds.gradient(np.ones(p), reps=40)
    out.append(ms)
print(os.environ.get("SLM_HIP_LIBRARY", "in-tree"), n, p, " ".join(f"{m:.4f}" for m in out), "ms", f"-> {8.0 * n * p / min(out) / 1e6:.0f} GB/s")
