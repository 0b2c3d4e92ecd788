#!/usr/bin/env python3
"""What a full-p Gram (covariance) mode would have to pay up front at the headline shape: X^T X in fp64 for
n = 100 000, p = 5 000 through the BLAS library of the box (torch -> rocBLAS/hipBLASLt), against the cost of the
passes over X it would replace ("dense solutions")."""
import time

import torch

n, p = 100_000, 5_000
X = torch.randn(n, p, dtype=torch.float64, device="cuda")
for _ in range(2):
    G = X.T @ X
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    G = X.T @ X
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"X^T X, fp64, n={n} p={p}: {1e3 * dt:.1f} ms = {2.0 * n * p * p / dt / 1e12:.1f} TFLOP/s (full product; a symmetric "
      f"update needs half the flops); one pass over X for sixteen problems costs 0.61 ms, two reads 1.37 ms")
Z = torch.randn(p, 16, dtype=torch.float64, device="cuda")
for _ in range(3):
    Y = G @ Z
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    Y = G @ Z
torch.cuda.synchronize()
print(f"G Z (p x p times p x 16): {1e6 * (time.perf_counter() - t0) / 20:.1f} us per product")
