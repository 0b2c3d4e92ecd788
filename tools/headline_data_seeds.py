#!/usr/bin/env python3
"""The headline path on OTHER draws of the headline's law (same coefficients, other X and noise): passes and ms per lane count
and working-set parameters -- is the three-pass path a property of the law or of the bench's one dataset?
usage: headline_data_seeds.py [sweep]"""
import itertools, os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import make_coef
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p, K = 100000, 5000, 50
seeds = (7, 1000, 1001, 1002, 1003, 1004, 1005, 1006)
sweep = len(sys.argv) > 1
combos = [(16, None, None, None), (18, None, None, None), (20, None, None, None)]
if sweep:
    for lanes in (18, 20):
        for kinit, app, theta in itertools.product((112, 144), (48, 72, 96), (0.85, 0.75, 0.65)):
            combos.append((lanes, kinit, app, theta))
tot = {c: [0, 0.0, []] for c in combos}
coef = make_coef(p, 50, seed=0)
for dseed in seeds:
    with eng.synthetic_dataset(n, p, seed=dseed, coef=coef, noise_sd=10.0) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
        for c in combos:
            lanes, kinit, app, theta = c
            for k, v in (("SLM_WS_KINIT", kinit), ("SLM_WS_APPEND", app), ("SLM_WS_THETA", theta)):
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = str(v)
            ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L)
            eng.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                r = ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L)
            eng.synchronize(); dt = (time.perf_counter() - t0) / 3
            tot[c][0] += r.grad_launches; tot[c][1] += dt; tot[c][2].append(int(r.grad_launches))
            assert r.converged
for c in combos:
    print(f"lanes={c[0]} kinit={c[1]} append={c[2]} theta={c[3]}: passes {tot[c][2]} = {tot[c][0]}, {1e3 * tot[c][1] / len(seeds):.3f} ms per path on average", flush=True)
