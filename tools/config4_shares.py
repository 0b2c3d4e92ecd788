"""BASELINE config 4 (2500 SparseGroupLasso fits) on one GPU and as the eight shares of an 8-rank job, each timed on
this GPU in turn: `bench.py`'s `config4_grid` / `config4_grid_emulated_world8` legs on their own.
Usage: python tools/config4_shares.py [n] [p] [emulated world]"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
import bench  # noqa: E402
from sparselm_amd import _engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000
w = int(sys.argv[3]) if len(sys.argv) > 3 else 8
eng = _engine.get_engine(0)
out = bench.leg_config4_grid(eng, 0, 1, n, p, emulate_world=w)
em = out.pop("emulated")
print(json.dumps(out))
worst = max(s["seconds"] for s in em["shares"])
for r, s in enumerate(em["shares"]):
    print(r, json.dumps(s))
print(f"full grid {out['seconds']:.4f} s / {out['passes']} passes; slowest share {worst:.4f} s; speed-up {out['seconds'] / worst:.2f}")
