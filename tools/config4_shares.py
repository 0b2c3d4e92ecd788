"""BASELINE config 4 (2500 SparseGroupLasso fits) on one GPU and as the eight shares of an 8-rank job, each timed on
this GPU in turn: `bench.py`'s `config4_grid` / `config4_grid_emulated_world8` legs on their own, with the plan's
options open.  Usage: python tools/config4_shares.py [world] [spread 1/0] [snake 1/0]"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
import bench  # noqa: E402
from sparselm_amd import _engine  # noqa: E402

w = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spread = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
snake = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
eng = _engine.get_engine(0)
c4 = bench.Config4(eng, 100_000, 5_000)
full = c4.calls_of(1, 0)
c4.run(full)
t1, p1 = min(c4.run(full) for _ in range(2))
print(f"full grid: {len(full)} calls, {p1} passes, {t1:.4f} s")
worst = 0.0
for r in range(w):
    calls = c4.calls_of(w, r, spread=spread, snake=snake)
    c4.run(calls)
    sec, pas = min(c4.run(calls) for _ in range(2))
    worst = max(worst, sec)
    print(r, json.dumps({"seconds": sec, "passes": pas, "lanes": [len(c) for c in calls],
                         "row_masks": [len({c4.units[u][0] for lane in c for u, _ in lane}) for c in calls]}))
print(f"spread={spread} snake={snake}: slowest share {worst:.4f} s; speed-up {t1 / worst:.2f}")
c4.close()
