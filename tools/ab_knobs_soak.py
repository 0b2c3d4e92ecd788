#!/usr/bin/env python3
"""A/B of environment knobs over the datasets of the soak law (bench.soak_case: 5-190 informative features, noise 0.1-100, path
floor 1e-3...0.1 alpha_max; n=100k, p=5k, 50 alphas, the engine's choice of lanes): per dataset and knob set the passes over X, the
light passes and the median ms; means over the sparse-ended and the dense-ended paths last.
usage: ab_knobs_soak.py N_DATASETS "" "SLM_NO_LAG_HANDOVER=1" ..."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import soak_case
from sparselm_amd import _engine
seeds = int(sys.argv[1])
sets = sys.argv[2:] or [""]
eng = _engine.get_engine(0)
n, p = 100000, 5000
def apply(setting):
    for k in [k for k in os.environ if k.startswith("SLM_")]:
        del os.environ[k]
    for kv in setting.split():
        k, v = kv.split("=", 1)
        os.environ[k] = v
rows = {s: [] for s in sets}
for seed in range(seeds):
    coef, noise, lo, k = soak_case(seed, p)
    with eng.synthetic_dataset(n, p, seed=100 + seed, coef=coef, noise_sd=noise) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, lo * amax, 50)]
        ref = None
        for s in sets:
            apply(s)
            for _ in range(2):
                r = ds.solve_path(pts, lanes=0, flags=_engine.FLAG_FRESH_L)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); r = ds.solve_path(pts, lanes=0, flags=_engine.FLAG_FRESH_L); ts.append(time.perf_counter() - t0)
            if ref is None:
                ref = r.betas.copy()
            dev = float(np.max(np.abs(r.betas - ref)) / max(np.max(np.abs(ref)), 1e-300))
            nnz = int(np.count_nonzero(r.betas[-1]))
            rows[s].append((1e3 * float(np.median(ts)), int(r.grad_launches), int(r.light_passes), nnz))
            print(f"seed {seed:2d} [{s or 'default'}]: {rows[s][-1][0]:7.3f} ms, {r.grad_launches:2d} passes + {r.light_passes} light, nnz_last {nnz}, converged {r.converged}, vs first set {dev:.1e}", flush=True)
apply("")
for s in sets:
    sp = [r for r in rows[s] if r[3] <= 512]
    de = [r for r in rows[s] if r[3] > 512]
    print(f"[{s or 'default'}] sparse-ended {len(sp)}: mean {np.mean([r[0] for r in sp]):.4f} ms, passes {sum(r[1] for r in sp)} + {sum(r[2] for r in sp)} light | dense-ended {len(de)}: mean {np.mean([r[0] for r in de]) if de else 0:.3f} ms, passes {sum(r[1] for r in de)}", flush=True)
