"""The reference's one real design matrix, and a strongly correlated full-size design, through the HIP engine.

`tests/golden/reference_examples/{corr,energy}.npy` are the data files of /root/reference/examples (290 cluster-expansion
correlation vectors x 66 features, condition number 650, and the energies they are fitted to): the only non-synthetic data the
reference ships.  /root/reference/examples/plot_chull.py:45-65 fits them with ``Lasso(fit_intercept=True, alpha=1.29e-5)``;
here the same fit, a 30-alpha path, a GroupLasso and an AdaptiveLasso fit run on the engine and are held against the oracle
(oracle/: numpy restatement of model/_lasso.py:99-121, 230-275 and model/_adaptive_lasso.py:158-232), scikit-learn's coordinate
descent and -- for the group penalty -- the independent active-set / Newton solver of tests/golden/second_solver.py, all to 1e-6
rel-inf (north star's bound).  Every other GPU test uses iid Gaussian or make_regression designs.

The last test is a 100 000 x 5 000 AR(1) design (rho = 0.9: eigenvalues of the population covariance from 0.05 to 19) with the
headline's coefficient law: its 50-alpha path is certified by the optimality conditions evaluated with the C twin's gradient
(oracle/fista_ref.c, plain C + OpenMP) and by the twin's fixed point, as tests/test_full_size_referee_gpu.py does for iid data.
"""

import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle
from oracle import cref
from sparselm_amd import _engine
from sparselm_amd.model import AdaptiveLasso, GroupLasso, Lasso

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import second_solver  # noqa: E402

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
for _v in ("OMP_NUM_THREADS",):
    os.environ.setdefault(_v, "16")


@pytest.fixture(scope="module")
def corr():
    d = os.path.join(HERE, "golden", "reference_examples")
    X, y = np.load(os.path.join(d, "corr.npy")), np.load(os.path.join(d, "energy.npy"))
    assert X.shape == (290, 66) and y.shape == (290,)
    return X, y


@pytest.fixture(scope="module")
def eng():
    return _engine.get_engine(0)


def _rel_inf(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def test_the_examples_own_lasso_fit(corr):
    """plot_chull.py:54,65 -- Lasso(fit_intercept=True, alpha=1.29e-5).fit(corr, energy)."""
    from sklearn.linear_model import Lasso as SkLasso

    X, y = corr
    est = Lasso(fit_intercept=True, alpha=1.29e-5).fit(X, y)
    ref = oracle.fit_lasso(X, y, alpha=1.29e-5, fit_intercept=True, tol=1e-14, max_iter=2_000_000)
    assert _rel_inf(est.coef_, ref["coef"]) <= 1e-6
    assert abs(est.intercept_ - ref["intercept"]) <= 1e-6 * max(1.0, abs(ref["intercept"]))
    sk = SkLasso(alpha=1.29e-5, fit_intercept=True, tol=1e-14, max_iter=5_000_000).fit(X, y)  # same objective, other algorithm
    assert _rel_inf(est.coef_, sk.coef_) <= 1e-6
    assert np.count_nonzero(est.coef_) == np.count_nonzero(np.abs(sk.coef_) > 1e-12)


def test_thirty_alpha_lasso_path_on_the_correlated_design(corr, eng):
    X, y = corr
    Xc, yc = X - X.mean(axis=0), y - y.mean()  # (the preprocessing of fit_intercept=True, model/_base.py:207-227)
    n, p = Xc.shape
    amax = float(np.max(np.abs(Xc.T @ yc)) / n)
    alphas = np.geomspace(amax, 1e-4 * amax, 30)
    gidx, G = oracle.group_index(None, p)
    with eng.dataset(Xc, yc) as ds:
        res = ds.solve_path([(a, 0.0, 0.0) for a in alphas], tol=1e-11)
        lanes = ds.solve_path([(a, 0.0, 0.0) for a in alphas], tol=1e-11, lanes=8, flags=_engine.FLAG_WORKING_SET)
    assert res.converged and lanes.converged
    b = None
    L = oracle.lipschitz(Xc)
    for k, a in enumerate(alphas):
        b, _ = oracle.fista(Xc, yc, a, 0.0, 0.0, gidx, G, beta0=b, L=L, tol=1e-14, max_iter=3_000_000)
        if np.max(np.abs(b)) > 0:
            assert _rel_inf(res.betas[k], b) <= 1e-6, (k, _rel_inf(res.betas[k], b))
            assert _rel_inf(lanes.betas[k], b) <= 1e-6, (k, _rel_inf(lanes.betas[k], b))
    assert np.count_nonzero(res.betas[-1]) >= 20  # (the path reaches its dense, ill-conditioned end)


def test_group_lasso_on_the_correlated_design(corr):
    X, y = corr
    groups = np.arange(66) // 6  # eleven groups of six neighbouring clusters
    alpha = 2e-3
    est = GroupLasso(groups=groups, alpha=alpha, fit_intercept=True).fit(X, y)
    ref = oracle.fit_group_lasso(X, y, groups=groups, alpha=alpha, fit_intercept=True, tol=1e-14, max_iter=3_000_000)
    assert _rel_inf(est.coef_, ref["coef"]) <= 1e-6
    # ... and the oracle itself against a solver that shares nothing with it (active set + Newton on the face)
    Xc, yc = X - X.mean(axis=0), y - y.mean()
    gidx, G = oracle.group_index(groups, 66)
    second = second_solver.solve(Xc, yc, np.zeros(66), alpha * np.ones(G), np.zeros(G), gidx, G)
    assert _rel_inf(est.coef_, second) <= 1e-6
    norms = np.sqrt(np.bincount(gidx, weights=est.coef_**2, minlength=G))
    assert 0 < np.count_nonzero(norms) < G  # some groups in, some out: all-or-nothing (tests/test_lasso.py:106-111)
    for g in range(G):
        assert np.all(est.coef_[gidx == g] != 0) or np.all(est.coef_[gidx == g] == 0)


def test_adaptive_lasso_on_the_correlated_design(corr):
    X, y = corr
    alpha = 1e-4
    est = AdaptiveLasso(alpha=alpha, fit_intercept=True, max_iter=3).fit(X, y)
    ref = oracle.fit_adaptive_lasso(X, y, alpha=alpha, fit_intercept=True, max_iter=3, tol=1e-10)
    assert est.n_iter_ == ref["n_iter"]
    assert _rel_inf(est.coef_, ref["coef"]) <= 1e-6
    plain = Lasso(alpha=alpha, fit_intercept=True).fit(X, y)
    assert np.count_nonzero(est.coef_) <= np.count_nonzero(plain.coef_)  # (tests/test_lasso.py:77-85)


def _ar1_design(n, p, rho, seed, threads=16):
    """X[i, j] = rho X[i, j-1] + sqrt(1 - rho^2) e[i, j], unit variance per column; rows generated by `threads` seeded streams."""
    from scipy.signal import lfilter

    X = np.empty((n, p))
    bounds = np.linspace(0, n, threads + 1).astype(int)
    seeds = np.random.SeedSequence(seed).spawn(threads)

    def fill(t):
        lo, hi = bounds[t], bounds[t + 1]
        e = np.random.default_rng(seeds[t]).standard_normal((hi - lo, p))
        e[:, 1:] *= np.sqrt(1.0 - rho * rho)
        X[lo:hi] = lfilter([1.0], [1.0, -rho], e, axis=1)

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(fill, range(threads)))
    return X


def test_ar1_correlated_full_size_path_is_certified_by_the_c_twin(eng):
    n, p, K, rho = 100_000, 5_000, 50, 0.9
    rng = np.random.default_rng(0)
    coef = np.zeros(p)
    coef[rng.choice(p, 50, replace=False)] = 100.0 * rng.uniform(size=50)
    X0 = _ar1_design(n, p, rho, seed=5)
    y = X0 @ coef + 10.0 * np.random.default_rng(6).standard_normal(n)
    with eng.dataset(X0, y) as ds:
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        alphas = np.geomspace(amax, 1e-3 * amax, K)
        res = ds.solve_path([(a, 0.0, 0.0) for a in alphas], lanes=0)
    assert res.converged
    with cref.NumaMatrix(X0) as X:
        del X0
        # power iteration for the twin's step size
        v = np.random.default_rng(0).standard_normal(p)
        lam = 1.0
        for _ in range(12):
            v /= np.linalg.norm(v)
            gv, _ = cref.gradient(X, 0.0 * y, v)
            lam = float(np.linalg.norm(gv))
            v = gv
        L = 1.1 * lam
        single = np.arange(p, dtype=np.int32)
        for k in (12, 30, 49):
            beta = res.betas[k]
            top = float(np.max(np.abs(beta)))
            # (i) optimality conditions with a gradient the engine did not compute, on the scale of the gradient at zero
            #     (a residual r bounds the distance to the minimiser by r / mu, mu >= (1 - rho) / (1 + rho) = 0.05 here)
            g, _ = cref.gradient(X, y, beta)
            on = beta != 0
            assert np.max(np.abs(g[~on])) <= alphas[k] + 1e-6 * amax, k
            kkt = float(np.max(np.abs(g[on] + alphas[k] * np.sign(beta[on]))))
            assert kkt <= 1e-6 * amax, (k, kkt / amax)
            # (ii) the twin, started at the engine's point, stays there
            b, _ = cref.fista(X, y, alphas[k], 0.0, 0.0, single, p, beta0=beta, L=L, tol=1e-10, max_iter=300)
            assert np.max(np.abs(b - beta)) <= 1e-6 * top, (k, float(np.max(np.abs(b - beta)) / top))
