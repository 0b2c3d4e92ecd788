#!/usr/bin/env python3
"""The headline path on 2 / 3 / 4 engines (streams) of one GPU at once (bench.leg_concurrent_paths)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd")); sys.path.insert(0, ROOT)
import bench
from sparselm_amd import _engine
eng = _engine.get_engine(0)
for s in (2, 3, 4):
    r = bench.leg_concurrent_paths(eng, 0, 0, 100000, 5000, 50, 1e-8, 16, streams=s, steps=10)
    print(s, round(r["fits_per_s_one_stream"]), round(r["fits_per_s_all_streams"]), r["converged"], flush=True)
