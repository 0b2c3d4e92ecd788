#!/usr/bin/env python3
"""A/B of environment knobs over DRAWS of the headline's law (bench.py leg_headline_draws' seeds + the bench's own dataset):
per knob set and draw the passes over X, the light passes and the median ms of a 50-alpha path; means over the draws last.
usage: ab_knobs_draws.py "" "SLM_SAMPLE_DIV=8" "SLM_AUTO_LANES=25" ... [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import make_coef
from sparselm_amd import _engine
sets = [a for a in sys.argv[1:] if not a.isdigit()] or [""]
reps = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 9
eng = _engine.get_engine(0)
n, p, K = 100000, 5000, 50
coef = make_coef(p, 50, seed=0)
seeds = (1000, 7, 1001, 1002, 1003, 1004, 1005, 1006, 1007)
rows = {s: [] for s in sets}
def apply(setting):
    for k in [k for k in os.environ if k.startswith("SLM_")]:
        del os.environ[k]
    for kv in setting.split():
        k, v = kv.split("=", 1)
        os.environ[k] = v
for dseed in seeds:
    with eng.synthetic_dataset(n, p, seed=dseed, coef=coef, noise_sd=10.0) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
        ref = None
        for s in sets:
            apply(s)
            for _ in range(2):
                r = ds.solve_path(pts, lanes=0, flags=_engine.FLAG_FRESH_L)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); r = ds.solve_path(pts, lanes=0, flags=_engine.FLAG_FRESH_L); ts.append(time.perf_counter() - t0)
            if ref is None:
                ref = r.betas.copy()
            dev = float(np.max(np.abs(r.betas - ref)) / np.max(np.abs(ref)))
            rows[s].append((1e3 * float(np.median(ts)), int(r.grad_launches), int(r.light_passes)))
            print(f"draw {dseed} [{s or 'default'}]: {rows[s][-1][0]:.3f} ms, {r.grad_launches} passes + {r.light_passes} light ({r.light_columns} columns), converged {r.converged}, rel-inf vs first set {dev:.1e}", flush=True)
apply("")
for s in sets:
    v = np.array([r[0] for r in rows[s]])
    print(f"[{s or 'default'}] mean {v.mean():.4f} ms (bench dataset {v[0]:.4f}, other draws {v[1:].mean():.4f}), passes {sum(r[1] for r in rows[s])}, light {sum(r[2] for r in rows[s])}", flush=True)
