"""Where the host's time goes in one share of config 4 (bench.Config4.run_call): building the lane specs, the call itself
(the engine's own wall inside it: SolveStats) -- usage: share_host_time.py [rank] [world] [contiguous]
(`contiguous`: the five folds as KFold(5) WITHOUT shuffling makes them -- row ranges -- instead of config 4's shuffled ones)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
import bench
from sparselm_amd import _engine
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 0
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
eng = _engine.get_engine(0)
c4 = bench.Config4(eng, 100_000, 5_000)
if len(sys.argv) > 3 and sys.argv[3] == "contiguous":
    import numpy as np
    fold = np.arange(100_000) * 5 // 100_000
    c4.masks = [(fold != f).astype(float) for f in range(5)]
calls = c4.calls_of(world, rank)
d = c4.ds
for rep in range(5):
    for call in calls:
        t0 = time.perf_counter()
        specs = []
        for lane in call:
            pts, gam = _engine.lane_points([c4.unit_pts[u][idx] for u, idx in lane])
            f = c4.units[lane[0][0]][0]
            specs.append(dict(points=pts, extrap=gam, row_weight=c4.masks[f], n_eff=int(c4.masks[f].sum())))
        t1 = time.perf_counter()
        out = d.solve_lanes(specs, flags=c4.flags)
        t2 = time.perf_counter()
        st = out[0]
        print(f"rep {rep}: specs {1e3 * (t1 - t0):.3f} ms, solve_lanes {1e3 * (t2 - t1):.3f} ms (engine wall {getattr(st, 'wall_ms', float('nan')):.3f}), passes {st.grad_launches}", flush=True)
c4.close()
