#!/usr/bin/env python3
"""BASELINE config 4 (2 500-fit SparseGroupLasso grid) on one GPU with its four 16-lane calls dealt to 1, 2 or 3
engines (streams) of the device, a dataset copy and a host thread each."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sparselm_amd import _engine

n, p = 100_000, 5_000
G = p // 10
rng = np.random.default_rng(1)
groups = rng.permutation(np.repeat(np.arange(G), 10))
coef = np.zeros(p)
for g in rng.choice(G, 25, replace=False):
    coef[groups == g] = 100.0 * rng.uniform(size=10)
engines = [_engine.get_engine(0)] + [_engine.Engine(0) for _ in range(2)]
sets = []
for e in engines:
    ds = e.synthetic_dataset(n, p, seed=11, coef=coef, noise_sd=10.0)
    ds.set_groups(groups, G)
    sets.append(ds)
g0, _ = sets[0].gradient(None)
gnorm = np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G))
bmax, amax1 = float(gnorm.max()), float(np.max(np.abs(g0)))
l1_ratios = np.linspace(0.05, 0.95, 10)
folds = np.random.default_rng(0).permutation(n) % 5
masks = [(folds != f).astype(float) for f in range(5)]
units = [(f, r) for f in range(5) for r in l1_ratios]
lanes = 16
batches = [units[k0:k0 + lanes] for k0 in range(0, len(units), lanes)]


def run_batch(ds, batch):
    split = max(1, lanes // len(batch))
    specs = []
    for f, r in batch:
        amax = min(bmax / (1 - r), amax1 / r)
        al = np.geomspace(amax, 1e-3 * amax, 50)
        pts = np.c_[r * al, (1 - r) * al, 0 * al]
        for part in np.array_split(np.arange(50), split):
            specs.append(dict(points=pts[part], row_weight=masks[f], n_eff=int(masks[f].sum())))
    out = ds.solve_lanes(specs)
    assert all(o.converged for o in out)
    return out[0].grad_launches


def run_grid(streams):
    todo = list(range(len(batches)))
    lock = threading.Lock()
    passes = [0]

    def work(i):
        while True:
            with lock:
                if not todo:
                    return
                b = todo.pop(0)
            k = run_batch(sets[i], batches[b])
            with lock:
                passes[0] += k

    threads = [threading.Thread(target=work, args=(i,)) for i in range(streams)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return time.perf_counter() - t0, passes[0]


for s in (1, 2, 3):
    run_grid(s)
for rep in range(2):
    for s in (1, 2, 3):
        dt, k = run_grid(s)
        print(f"{s} stream(s): {dt:.4f} s per grid = {2500 / dt:.0f} fits/s, {k} passes", flush=True)
