#!/usr/bin/env python3
"""A/B of environment knobs on ONE box, ONE process: the headline path (and config 3's group path) timed with each knob set in
turn, alternating.  usage: ab_knobs.py "" SLM_NO_SAMPLE_START=1 "SLM_WS_KINIT=144 SLM_WS_APPEND=64" ... [rounds]"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import make_coef
from sparselm_amd import _engine
sets = [a for a in sys.argv[1:] if not a.isdigit()] or [""]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
eng = _engine.get_engine(0)
n, p, K = 100000, 5000, 50
coef = make_coef(p, 50, seed=0)
res = {s: {"headline": [], "group": []} for s in sets}
passes = {}
with eng.synthetic_dataset(n, p, seed=1000, coef=coef, noise_sd=10.0) as ds:
    g0, _ = ds.gradient(None)
    amax = float(np.max(np.abs(g0)))
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
    groups = np.random.default_rng(1).permutation(np.repeat(np.arange(500), 10)).astype(np.int32)
    bmax = float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=500))))
    pts3 = [(0.0, b, 0.0) for b in np.geomspace(bmax, 1e-3 * bmax, K)]
    def apply(setting):
        for k in [k for k in os.environ if k.startswith("SLM_")]:
            del os.environ[k]
        for kv in setting.split():
            k, v = kv.split("=", 1)
            os.environ[k] = v
    def timed(points, reps):
        for _ in range(3):
            r = ds.solve_path(points, lanes=0, flags=_engine.FLAG_FRESH_L)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); r = ds.solve_path(points, lanes=0, flags=_engine.FLAG_FRESH_L); ts.append(time.perf_counter() - t0)
        return 1e3 * float(np.median(ts)), int(r.grad_launches)
    for rnd in range(rounds):
        ds.set_groups(None)
        for s in sets:
            apply(s)
            ms, ps = timed(pts, 40)
            res[s]["headline"].append(ms); passes[(s, "headline")] = ps
        ds.set_groups(groups, 500)
        for s in sets:
            apply(s)
            ms, ps = timed(pts3, 20)
            res[s]["group"].append(ms); passes[(s, "group")] = ps
for s in sets:
    for what in ("headline", "group"):
        v = res[s][what]
        print(f"[{s or 'default'}] {what}: median {np.median(v):.4f} ms, {passes[(s, what)]} passes, all {['%.4f' % x for x in v]}", flush=True)
