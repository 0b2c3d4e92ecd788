"""A/B of two builds of the engine on ONE box: the headline path (and config 3's group path) timed in child processes
that load the library named by SLM_HIP_LIBRARY, alternating.  usage: ab_headline.py libA.so libB.so [...] [rounds]"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

CHILD = r'''
import os, sys, time, json
import numpy as np
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
from bench import make_coef
from sparselm_amd import _engine
eng = _engine.get_engine(0)
n, p, K = 100000, 5000, 50
out = {}
coef = make_coef(p, 50, seed=0)
with eng.synthetic_dataset(n, p, seed=1000, coef=coef, noise_sd=10.0) as ds:
    g0, _ = ds.gradient(None)
    amax = float(np.max(np.abs(g0)))
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 1e-3 * amax, K)]
    for _ in range(5):
        r = ds.solve_path(pts, lanes=0, flags=_engine.FLAG_FRESH_L)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); r = ds.solve_path(pts, lanes=0, flags=_engine.FLAG_FRESH_L); ts.append(time.perf_counter() - t0)
    out["headline_ms"] = 1e3 * float(np.median(ts)); out["headline_passes"] = int(r.grad_launches)
    # config 3: 500 groups of 10, group lasso path
    groups = np.arange(p) // 10
    ds.set_groups(groups, 500)
    gn = np.sqrt(np.bincount(groups, weights=g0 ** 2))
    bmax = float(np.max(gn / np.sqrt(10.0)))
    pts3 = [(0.0, b, 0.0) for b in np.geomspace(bmax, 1e-2 * bmax, K)]
    try:
        for _ in range(3):
            r = ds.solve_path(pts3, lanes=0, flags=_engine.FLAG_FRESH_L)
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); r = ds.solve_path(pts3, lanes=0, flags=_engine.FLAG_FRESH_L); ts.append(time.perf_counter() - t0)
        out["group_ms"] = 1e3 * float(np.median(ts)); out["group_passes"] = int(r.grad_launches)
    except Exception as e:
        out["group_err"] = repr(e)[:200]
print(json.dumps(out))
'''

def run(lib):
    env = dict(os.environ, SLM_HIP_LIBRARY=os.path.abspath(lib))
    o = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True)
    if o.returncode != 0:
        print(o.stderr[-2000:]); raise SystemExit(1)
    return json.loads(o.stdout.strip().splitlines()[-1])

if __name__ == "__main__":
    libs = [x for x in sys.argv[1:] if not x.isdigit()]
    rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
    res = {lib: [] for lib in libs}
    for _ in range(rounds):
        for lib in libs:
            r = run(lib); res[lib].append(r); print(os.path.basename(lib), json.dumps(r), flush=True)
    for lib in libs:
        for key in ("headline_ms", "group_ms"):
            v = [r[key] for r in res[lib] if key in r]
            if v: print(f"{os.path.basename(lib):>20} {key}: median {np.median(v):.4f}  all {['%.4f' % x for x in v]}")
