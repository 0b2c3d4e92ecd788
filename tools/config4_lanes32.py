#!/usr/bin/env python3
"""BASELINE config 4's grid over X with sixteen and with thirty-two lanes per call (two halves on one read of X)."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
import bench
from sparselm_amd import _engine
eng = _engine.get_engine(0)
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
c4 = bench.Config4(eng, 100_000, 5_000, noise_sd=noise)
ref = {}
for lanes in (16, 32, 16, 32):
    c4.lanes = lanes
    calls = c4.calls_of(1, 0)
    keep = {}
    for call in calls:
        c4.run_call(c4.ds, call, keep)
    sec, passes = min(c4.run(calls) for _ in range(2))
    if not ref:
        ref = keep
    worst = max(float(np.max(np.abs(keep[k] - ref[k])) / max(np.max(np.abs(ref[k])), 1e-300)) for k in ref)
    print(f"config 4 (noise {noise:g}) over X, {lanes} lanes per call: {sec:.4f} s per 2500-fit grid, {passes} passes in {len(calls)} calls ({[len(c) for c in calls]} lanes), vs the first run {worst:.1e}", flush=True)
c4.close()
