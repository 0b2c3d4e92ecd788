#!/usr/bin/env python3
"""The soak seeds of the headline shape on several lane counts: ms and passes per path.  usage: lanes_soak.py lanes... -- seeds..."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import soak_case
from sparselm_amd import _engine
eng = _engine.get_engine(0)
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
lane_counts = [int(a) for a in args[:cut]] or [16, 18, 20]
seeds = [int(a) for a in args[cut + 1:]] or list(range(4, 16))
n, p, K = 100000, 5000, 50
tot = {l: 0.0 for l in lane_counts}
for seed in seeds:
    coef, noise, lo, k = soak_case(seed, p)
    with eng.synthetic_dataset(n, p, seed=100 + seed, coef=coef, noise_sd=noise) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, lo * amax, K)]
        row, ref = [], None
        for lanes in lane_counts:
            ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L)
            best = 1e9
            for _ in range(2):
                eng.synchronize(); t0 = time.perf_counter()
                r = ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L)
                best = min(best, time.perf_counter() - t0)
            if ref is None: ref = r.betas
            err = float(np.max(np.abs(r.betas - ref)) / np.max(np.abs(ref)))
            tot[lanes] += best
            row.append(f"{lanes}: {1e3*best:6.2f} ms/{r.grad_launches:2d}p{'' if r.converged and err < 1e-6 else ' CHECK'}")
        print(f"seed {seed:2d} nnz_last {np.count_nonzero(ref[-1]):4d}  " + "  ".join(row), flush=True)
print("total ms: " + "  ".join(f"{l}: {1e3*t:.1f}" for l, t in tot.items()))
