"""The reference's solve has no width or group-size limit (/root/reference/src/sparselm/model/_base.py:512-519: any p;
model/_lasso.py:239-255: groups of any size).  Until round 6 the engine had three: rows beyond 10 240 columns ran one lane on
the two-pass kernels, ONE group of more than 64 features switched the working set off for the whole dataset, and the model
Gram stops at 16 384 columns.  Here: sixteen lanes and the working set at p = 20 000 (groups of ten), a 200-feature group
among small ones, both against the oracle to 1e-6 -- and the model Gram's bound as an explicit refusal that leaves the solve
itself intact.
"""

import os

import numpy as np
import pytest

import oracle
from oracle import cref
from sparselm_amd import _engine

pytestmark = pytest.mark.gpu

for _v in ("OMP_NUM_THREADS",):
    os.environ.setdefault(_v, "16")


@pytest.fixture(scope="module")
def eng():
    return _engine.get_engine(0)


def test_sixteen_lanes_and_the_working_set_at_twenty_thousand_columns(eng):
    n, p, gsize = 4096, 20_000, 10
    G = p // gsize
    rng = np.random.default_rng(12)
    X = rng.standard_normal((n, p))
    groups = rng.permutation(np.repeat(np.arange(G), gsize)).astype(np.int32)
    beta = np.zeros(p)
    for g in rng.choice(G, 12, replace=False):
        beta[groups == g] = rng.uniform(2.0, 6.0, gsize) * rng.choice([-1.0, 1.0], gsize)
    y = X @ beta + rng.standard_normal(n)
    g0 = X.T @ y / n
    bmax = float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G))))
    K = 16
    balphas = np.geomspace(0.9 * bmax, 0.25 * bmax, K)
    with eng.dataset(X, y) as ds:
        ds.set_groups(groups, G)
        assert ds.max_lanes(0) == 16  # (the split pass serves this width: until round 6 one lane on the two-pass kernels)
        res = ds.solve_path([(0.0, b, 0.0) for b in balphas], lanes=16, tol=1e-9)
        one = ds.solve_path([(0.0, b, 0.0) for b in balphas], lanes=1, tol=1e-9)
    assert res.converged and one.converged
    assert res.ws_refined > 0 and res.grad_launches < one.grad_launches  # the working set, sixteen points per pass
    single = np.arange(p, dtype=np.int32)
    del single
    gidx, Gn = oracle.group_index(groups, p)
    L = 1.05 * float(np.linalg.norm(X, 2) ** 2) / n
    for k in (3, 9, 15):
        b, _ = cref.fista(X, y, 0.0, balphas[k], 0.0, gidx.astype(np.int32), Gn, beta0=res.betas[k], L=L, tol=1e-12, max_iter=20000)
        top = float(np.max(np.abs(b)))
        assert top > 0
        assert np.max(np.abs(res.betas[k] - b)) <= 1e-6 * top, (k, float(np.max(np.abs(res.betas[k] - b)) / top))
        assert np.max(np.abs(one.betas[k] - b)) <= 1e-6 * top
    # the gradient kernels of this width against numpy (split pass: residuals from the column-major copy, X^T R in 40 column blocks)
    z = np.zeros(p)
    z[rng.choice(p, 200, replace=False)] = rng.standard_normal(200)
    with eng.dataset(X, y) as ds:
        for lanes, lane in ((1, 0), (16, 7)):
            g, loss = ds.gradient(z, split=True, lanes=lanes, lane=lane)
            gr = X.T @ (X @ z - y) / n
            assert np.max(np.abs(g - gr)) <= 1e-12 * np.max(np.abs(gr))


def test_a_group_of_two_hundred_features_keeps_the_working_set(eng):
    n, p = 3000, 1000
    sizes = [200] + [10] * 50 + [30] * 10  # 200 + 500 + 300
    labels = np.repeat(np.arange(len(sizes)), sizes)
    rng = np.random.default_rng(21)
    groups = labels[rng.permutation(p)].astype(np.int32)
    G = len(sizes)
    X = rng.standard_normal((n, p))
    beta = np.zeros(p)
    beta[groups == 0] = rng.uniform(0.3, 1.0, 200) * rng.choice([-1.0, 1.0], 200)  # the big group is informative
    for g in (3, 17, 55):
        beta[groups == g] = rng.uniform(1.0, 3.0, int(np.sum(groups == g)))
    y = X @ beta + rng.standard_normal(n)
    g0 = X.T @ y / n
    w = np.sqrt(np.bincount(groups, minlength=G).astype(float))  # sqrt(size) weights
    bmax = float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G)) / w))
    balphas = np.geomspace(bmax, 0.05 * bmax, 12)
    with eng.dataset(X, y) as ds:
        ds.set_groups(groups, G)
        res = ds.solve_path([(0.0, b, 0.0) for b in balphas], b=w, lanes=4, tol=1e-10, flags=_engine.FLAG_WORKING_SET)
        plain = ds.solve_path([(0.0, b, 0.0) for b in balphas], b=w, lanes=4, tol=1e-10, flags=_engine.FLAG_NO_WORKING_SET)
    assert res.converged and plain.converged
    assert res.ws_refined > 0, "a group of more than 64 features switched the working set off"
    gidx, Gn = oracle.group_index(groups, p)
    b = None
    for k, a in enumerate(balphas):
        b, _ = oracle.fista(X, y, 0.0, a * w, 0.0, gidx, Gn, beta0=b, tol=1e-13)
        top = float(np.max(np.abs(b)))
        if top > 0:
            assert np.max(np.abs(res.betas[k] - b)) <= 1e-6 * top, (k, float(np.max(np.abs(res.betas[k] - b)) / top))
            assert np.max(np.abs(plain.betas[k] - b)) <= 1e-6 * top
    assert np.any(res.betas[-1][groups == 0] != 0)  # the big group is in the model at the end of the path


def test_the_model_gram_refuses_beyond_16384_columns_and_the_solve_goes_on(eng):
    n, p = 256, 16_500
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, p))
    y = X[:, :5] @ np.array([3.0, -2.0, 2.0, 1.5, -1.0]) + 0.1 * rng.standard_normal(n)
    with eng.dataset(X, y) as ds:
        with pytest.raises(NotImplementedError, match="model Gram"):  # (SLM_ERR_UNSUPPORTED)
            ds.model_gram()
        amax = float(np.max(np.abs(X.T @ y)) / n)
        res = ds.solve_path([(0.5 * amax, 0.0, 0.0), (0.3 * amax, 0.0, 0.0)], lanes=2, tol=1e-9)
    assert res.converged and res.mg_rounds == 0
    gidx, G = oracle.group_index(None, p)
    b, _ = oracle.fista(X, y, 0.3 * amax, 0.0, 0.0, gidx, G, tol=1e-13)
    assert np.max(np.abs(res.betas[1] - b)) <= 1e-6 * np.max(np.abs(b))
