"""What the pieces of a row-sharded Gram build would cost for BASELINE config 4 at eight ranks, measured with the code
that exists: (a) every rank's share solved from the (complete) fold Grams, (b) the Grams of an eighth of the rows.
Usage: python tools/config4_sharded_probe.py [world]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
import bench  # noqa: E402
from sparselm_amd import _engine  # noqa: E402

w = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = _engine.get_engine(0)
n, p = 100_000, 5_000
c4 = bench.Config4(eng, n, p)
full = c4.calls_of(1, 0)
c4.run(full)
t1, p1 = min(c4.run(full) for _ in range(2))
print(f"one GPU over X: {len(full)} calls, {p1} passes, {t1:.4f} s")
shares_x = []
for r in range(w):
    calls = c4.calls_of(w, r)
    c4.run(calls)
    shares_x.append(min(c4.run(calls) for _ in range(2)))
print("shares over X:", json.dumps([round(1e3 * s, 2) for s, _ in shares_x]), [q for _, q in shares_x])
build = c4.build_covariance()
c4.run(full)
tc, pc = min(c4.run(full) for _ in range(2))
print(f"one GPU from Grams: build {build:.4f} s + {tc:.4f} s ({pc} passes)")
shares_c = []
for r in range(w):
    calls = c4.calls_of(w, r)
    c4.run(calls)
    shares_c.append(min(c4.run(calls) for _ in range(3)))
print("shares from Grams:", json.dumps([round(1e3 * s, 2) for s, _ in shares_c]), [q for _, q in shares_c])
c4.close()

# (b) the Grams of one rank's rows: a dataset of n / w rows with the same fold structure
m = n // w
coef = np.zeros(p)
coef[:50] = 1.0
ds = eng.synthetic_dataset(m, p, seed=3, coef=coef, noise_sd=1.0)
folds = np.random.default_rng(0).permutation(m) % 5
masks = [(folds != f).astype(float) for f in range(5)]
for rep in range(3):
    ds.center()  # (drops the Grams)
    eng.synchronize()
    t0 = time.perf_counter()
    ds.covariance_folds(masks, [int(mk.sum()) for mk in masks])
    eng.synchronize()
    print(f"Grams of {m} rows, five folds: {1e3 * (time.perf_counter() - t0):.2f} ms")
ds.close()
worst_c = max(s for s, _ in shares_c)
print(f"slowest share from Grams {1e3 * worst_c:.2f} ms; over X {1e3 * max(s for s, _ in shares_x):.2f} ms")
