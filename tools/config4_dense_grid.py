"""BASELINE config 4 with noise 100 instead of 10: the ends of its paths outgrow the 512-column working set, the grid's
dense regime.  What the grid costs today (plain sixteen-lane passes where the working set gives up), and what a
covariance route would cost from its measured parts: one Gram per fold (the whole X^T X once, minus each fold's test
block) shared by the ten l1_ratio rows, then G_f Z for sixteen lanes per iteration instead of two reads of X.
`python tools/config4_dense_grid.py [noise_sd]`"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
import bench  # noqa: E402
from sparselm_amd import _engine  # noqa: E402

noise = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
n, p = 100_000, 5_000
eng = _engine.get_engine(0)
c4 = bench.Config4(eng, n, p, noise_sd=noise)
calls = c4.calls_of(1, 0)
keep = {}
t0 = time.perf_counter()
passes = [c4.run_call(c4.ds, call, keep) for call in calls]
warm = time.perf_counter() - t0
t0 = time.perf_counter()
passes = [c4.run_call(c4.ds, call) for call in calls]
dt = time.perf_counter() - t0
nnz = np.array([np.count_nonzero(keep[(u, c4.K - 1)]) for u in range(len(c4.units))])
over = np.array([sum(np.count_nonzero(keep[(u, k)]) > 512 for k in range(c4.K)) for u in range(len(c4.units))])
print(f"config 4, noise {noise:g}: {dt:.3f} s per 2500-fit grid ({warm:.3f} s the first time), passes per call {passes} = {sum(passes)} in all; "
      f"non-zeros at the last alpha {nnz.min()}..{nnz.max()} (median {int(np.median(nnz))}); path points above 512 non-zeros: "
      f"{int(over.sum())} of 2500, in {int((over > 0).sum())} of 50 units", flush=True)
# the same grid from the Grams of its five folds
keep_x = keep
t0 = time.perf_counter()
t_build = c4.build_covariance()
keep_c = {}
passes_c = [c4.run_call(c4.ds, call, keep_c) for call in calls]
t0 = time.perf_counter()
passes_c = [c4.run_call(c4.ds, call) for call in calls]
dt_c = time.perf_counter() - t0
worst = max(float(np.max(np.abs(keep_c[k] - keep_x[k])) / max(np.max(np.abs(keep_x[k])), 1e-300)) for k in keep_x)
print(f"  with covariance passes (SLM_FLAG_COVARIANCE): five fold Grams built in {t_build:.3f} s, then {dt_c:.3f} s per grid, passes per call {passes_c} = {sum(passes_c)}; worst difference of a coefficient vector "
      f"to the run over X {worst:.2e} (relative to its largest entry)", flush=True)
c4.close()

# the parts of a covariance route, measured on this box
import torch  # noqa: E402

X = torch.randn(n, p, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
t_full = timed(lambda: X.T @ X)
blk = X[: n // 5]
t_blk = timed(lambda: blk.T @ blk)
G = X.T @ X
Z = torch.randn(p, 16, dtype=torch.float64, device="cuda")
t_gz = timed(lambda: G @ Z, 20)
plain = sum(passes)
print(f"covariance route, measured parts: X^T X {1e3 * t_full:.1f} ms, a fold's test block {1e3 * t_blk:.1f} ms -> five fold Grams "
      f"{1e3 * (t_full + 5 * t_blk):.1f} ms (1.0 GB); G Z for sixteen lanes {1e6 * t_gz:.0f} us by the BLAS library "
      f"(200 MB per product: {200e6 / 6.5e12 * 1e6:.0f} us at the rate of the pass kernels)")
print(f"  the grid's {plain} passes as iterations on the Grams: {1e3 * (t_full + 5 * t_blk) + plain * (1e3 * t_gz + 0.03):.0f} ms with the library product, "
      f"{1e3 * (t_full + 5 * t_blk) + plain * (0.031 + 0.03):.0f} ms with a product at the pass kernels' rate, against {1e3 * dt:.0f} ms today")
