"""Full-size results certified by something that is NOT the engine (round-4 verdict, "missing" item 5).

The KKT tests of tests/test_baseline_configs_gpu.py certify coefficients with a gradient the engine itself computes.  Here the
synthetic data are downloaded to the host and the oracle's C twin (oracle/fista_ref.c through oracle/cref.py: plain C + OpenMP,
its own power iteration for the step size; test infrastructure) is the judge:

 (i)  the gradient kernels at the shapes whose launch geometry no small test reaches -- 100 000 x 5 000 (one workgroup per CU
      walking ~390 rows, XCD tile remap; the fused kernel and both matrix-core halves of the split pass) and 125 000 x 10 000
      (BASELINE config 5's per-rank share: `grad_fused_kernel<8,10,1,1>`) -- against the twin's gradient to 1e-12;
 (ii) points of BASELINE config 2's Lasso path and of config 3's GroupLasso path at n = 100k, p = 5k: the twin, warm-started at
      the GPU's solution and run to 1e-10, may not move it by more than 1e-6 rel-inf (north star's bound) -- a solution that is
      not the minimiser would be carried away from where it stands.

Reference: the objective of /root/reference/src/sparselm/model/_lasso.py:99-121, 230-275 as restated in oracle/penalty.py."""

import os

import numpy as np
import pytest

from oracle import cref
from sparselm_amd import _engine

pytestmark = pytest.mark.gpu

# (the referee reads 4 GB per gradient on the host: bound its threads the way the smoke check does)
for _v in ("OMP_NUM_THREADS",):
    os.environ.setdefault(_v, "16")


@pytest.fixture(scope="module")
def eng():
    return _engine.get_engine(0)


def _twin_L(X, y, p):
    v = np.random.default_rng(0).standard_normal(p)
    lam = 1.0
    for _ in range(8):
        v /= np.linalg.norm(v)
        gv, _ = cref.gradient(X, 0.0 * y, v)
        lam = float(np.linalg.norm(gv))
        v = gv
    return 1.1 * lam


@pytest.mark.parametrize("n,p", [(100_000, 5_000), (125_000, 10_000)])
def test_gradient_at_full_size_against_the_c_twin(eng, monkeypatch, n, p):
    rng = np.random.default_rng(3)
    coef = np.zeros(p)
    coef[rng.choice(p, 40, replace=False)] = 10.0 * rng.standard_normal(40)
    with eng.synthetic_dataset(n, p, seed=21, coef=coef, noise_sd=5.0) as ds:
        X, y = ds.download()
        z = np.zeros(p)
        z[rng.choice(p, 300, replace=False)] = rng.standard_normal(300)
        zd = rng.standard_normal(p) * 0.1  # (a dense point: every column takes part)
        for point in (None, z, zd):
            g, loss = ds.gradient(point)
            gr, lr = cref.gradient(X, y, np.zeros(p) if point is None else point)
            scale = float(np.max(np.abs(gr)))
            assert np.max(np.abs(g - gr)) <= 1e-12 * scale, (n, p, float(np.max(np.abs(g - gr)) / scale))
            assert abs(loss - lr) <= 1e-12 * abs(lr)
        if p <= 5_000:  # the split pass (rowdot_mfma_kernel + xtr_mfma_kernel: the sixteen-lane route of large X)
            g, loss = ds.gradient(zd, split=True)
            gr, lr = cref.gradient(X, y, zd)
            assert np.max(np.abs(g - gr)) <= 1e-12 * float(np.max(np.abs(gr)))
            assert abs(loss - lr) <= 1e-12 * abs(lr)


def test_config2_and_config3_solutions_are_fixed_points_of_the_c_twin(eng):
    n, p, K, G = 100_000, 5_000, 50, 500
    rng = np.random.default_rng(0)
    coef = np.zeros(p)
    coef[rng.choice(p, 50, replace=False)] = 100.0 * rng.uniform(size=50)
    groups = np.random.default_rng(1).permutation(np.repeat(np.arange(G), 10)).astype(np.int32)
    with eng.synthetic_dataset(n, p, seed=7, coef=coef, noise_sd=10.0) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        alphas = np.geomspace(amax, 1e-3 * amax, K)
        lasso = ds.solve_path([(a, 0.0, 0.0) for a in alphas], lanes=16)
        ds.set_groups(groups, G)
        bmax = float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G))))
        balphas = np.geomspace(bmax, 1e-3 * bmax, K)
        glasso = ds.solve_path([(0.0, a, 0.0) for a in balphas], lanes=16)
        assert lasso.converged and glasso.converged
        X0, y = ds.download()
    with cref.NumaMatrix(X0) as X:
        del X0
        L = _twin_L(X, y, p)
        single = np.arange(p, dtype=np.int32)
        for k in (10, 30, 49):  # config 2: top of the path, middle, end (~200 non-zeros)
            b, _ = cref.fista(X, y, alphas[k], 0.0, 0.0, single, p, beta0=lasso.betas[k], L=L, tol=1e-10, max_iter=400)
            top = float(np.max(np.abs(b)))
            assert np.max(np.abs(b - lasso.betas[k])) <= 1e-6 * top, (k, float(np.max(np.abs(b - lasso.betas[k])) / top))
        for k in (20, 49):  # config 3
            b, _ = cref.fista(X, y, 0.0, balphas[k], 0.0, groups, G, beta0=glasso.betas[k], L=L, tol=1e-10, max_iter=400)
            top = float(np.max(np.abs(b)))
            assert np.max(np.abs(b - glasso.betas[k])) <= 1e-6 * top, (k, float(np.max(np.abs(b - glasso.betas[k])) / top))
