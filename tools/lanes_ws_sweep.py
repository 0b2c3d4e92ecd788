#!/usr/bin/env python3
"""The headline path on wide lane counts against the working set's first size / append size / theta:
passes and ms per path.  usage: lanes_ws_sweep.py [seed ...]  (seed 0 = the bench's dataset, others = soak cases)"""
import itertools, os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import make_coef, soak_case
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p, K = 100000, 5000, 50
seeds = [int(a) for a in sys.argv[1:]] or [0]
for seed in seeds:
    if seed == 0:
        coef, noise, lo, dseed = make_coef(p, 50, seed=0), 10.0, 1e-3, 1000
    else:
        coef, noise, lo, _k = soak_case(seed, p); dseed = 100 + seed
    with eng.synthetic_dataset(n, p, seed=dseed, coef=coef, noise_sd=noise) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, lo * amax, K)]
        ref = None
        combos = [(16, None, None, None)]
        for lanes in (25, 26, 32):
            for kinit, app, theta in itertools.product((112, 160, 224, 288), (48, 96, 160), (0.85, 0.7)):
                combos.append((lanes, kinit, app, theta))
        for lanes, kinit, app, theta in combos:
            for k, v in (("SLM_WS_KINIT", kinit), ("SLM_WS_APPEND", app), ("SLM_WS_THETA", theta)):
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = str(v)
            ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L)
            eng.synchronize(); t0 = time.perf_counter()
            for _ in range(4):
                r = ds.solve_path(pts, lanes=lanes, flags=_engine.FLAG_FRESH_L)
            eng.synchronize(); dt = (time.perf_counter() - t0) / 4
            if ref is None: ref = r.betas.copy()
            err = float(np.max(np.abs(r.betas - ref)) / np.max(np.abs(ref)))
            print(f"seed {seed} lanes={lanes} kinit={kinit} append={app} theta={theta}: {1e3*dt:.3f} ms, {r.grad_launches} passes, "
                  f"ws b/a/m/cols {r.ws_builds}/{r.ws_appends}/{r.ws_misses}/{r.ws_columns}, conv {r.converged}, err {err:.1e}", flush=True)
