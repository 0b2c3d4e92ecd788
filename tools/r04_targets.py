"""The kernels of rounds 3-4 that the headline bench does not reach, one after the other, for `rocprofv3` (kernel trace or
counters; run the program directly after `--`): the folds' Grams of BASELINE config 4 (cov_syrk_packed_kernel, cov_unpack_kernel;
then one row set on its own: cov_syrk_kernel), config 4's grid from the Grams (cov_gz_mfma_kernel in both of its ways:
listed rows of the Gram, all rows), the reference's problem sizes on chip (small_solve_kernel, small_stdsgl_kernel), and
BASELINE config 5's per-rank shape (grad_fused_kernel<8,10,1,1> at p = 10 000, the streaming tail kernel).
usage: python tools/r04_targets.py [grams] [grid] [small] [config5]   (default: all)"""
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "sparse-lm_amd")]
import bench  # noqa: E402
from sparselm_amd import _engine  # noqa: E402

want = set(sys.argv[1:]) or {"grams", "grid", "small", "config5"}
eng = _engine.get_engine(0)
n, p = 100_000, 5_000


def note(msg):
    print(msg, file=sys.stderr, flush=True)


if want & {"grams", "grid"}:
    c4 = bench.Config4(eng, n, p)
    n_effs = [int(m.sum()) for m in c4.masks]
    for rep in range(3):
        c4.ds.covariance_clear()
        eng.synchronize()
        t0 = time.perf_counter()
        c4.ds.covariance_folds(c4.masks, n_effs)
        eng.synchronize()
        note(f"five folds' Grams: {1e3 * (time.perf_counter() - t0):.2f} ms")
    if "grams" in want:
        w = (np.random.default_rng(5).permutation(n) % 3 != 0).astype(float)  # a third left out: all rows minus those
        t0 = time.perf_counter()
        c4.ds.covariance(w, int(w.sum()))
        eng.synchronize()
        note(f"one more row set (33 333 rows left out; its rows in chunks through cov_syrk_packed_kernel): {1e3 * (time.perf_counter() - t0):.2f} ms")
    if "grid" in want:
        c4.flags |= _engine.FLAG_COVARIANCE
        calls = c4.calls_of(1, 0)
        for rep in range(3):
            sec, passes = c4.run(calls)
            note(f"config 4 from the Grams: {passes} passes, {1e3 * sec:.2f} ms")
        os.environ["SLM_COV_ALL_ROWS"] = "1"  # every pass reads all rows of its Grams
        for rep in range(2):
            sec, passes = c4.run(calls)
            note(f"... every pass over all rows of the Gram: {passes} passes, {1e3 * sec:.2f} ms")
        del os.environ["SLM_COV_ALL_ROWS"]
    c4.close()

if "small" in want:
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = bench.leg_config1_small()
    note("reference-sized problems: " + ", ".join(f"{k} = {v:.3f}" if isinstance(v, float) else f"{k} = {v}" for k, v in out.items()))

if "config5" in want:
    G = 1000
    groups = np.repeat(np.arange(G), 10)
    rng = np.random.default_rng(0)
    coef = np.zeros(10_000)
    for g in rng.choice(G, 30, replace=False):
        coef[groups == g] = rng.uniform(1, 5, 10)
    with eng.synthetic_dataset(125_000, 10_000, seed=1000, coef=coef, noise_sd=5.0) as ds:
        ds.set_groups(groups, G)
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G))))
        alpha, eps = 0.1 * amax, 1e-6
        for rep in range(3):
            w, beta, passes = alpha * np.ones(G), None, 0
            eng.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                res = ds.solve_path([(0.0, 1.0, 0.0)], b=w, beta0=beta, tol=1e-8, want_group_norms=True)
                beta = res.betas[0]
                passes += res.grad_launches
                w = alpha * (alpha / (res.group_norms[0] + eps))
            note(f"config 5's per-rank share (no communicator): {passes} passes, {1e3 * (time.perf_counter() - t0):.2f} ms per fit")
