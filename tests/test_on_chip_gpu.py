"""The on-chip solver for problems that fit a workgroup (csrc/small_kernels.hpp; SLM_FLAG_ON_CHIP): the sizes the
reference's own tests and README run (/root/reference/tests/conftest.py:17-19, README.md:42-55)."""

import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from sparselm_amd import _engine

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def eng():
    return _engine.get_engine(0)


def test_reference_sized_fits_match_the_oracle(eng):
    """25 x 20 and 25 x 30 (p > n) with the reference fixture's law, 100 x 80 as in its README: Lasso, group, sparse
    group and ridged penalties through ONE launch each, against the oracle's proximal-gradient iteration."""
    rng = np.random.default_rng(0)
    for n, p, G in ((25, 20, 4), (25, 30, 6), (100, 80, 8)):
        X = rng.standard_normal((n, p))
        beta = np.zeros(p)
        beta[rng.choice(p, 10, replace=False)] = rng.uniform(1, 5, 10)
        y = X @ beta + 0.5 * rng.standard_normal(n) + 3.0
        groups = rng.permutation(np.arange(p) % G)
        with eng.dataset(X, y) as ds:
            for kind, pen in (("lasso", (0.3, 0.0, 0.0)), ("group", (0.0, 0.6, 0.0)), ("sgl", (0.2, 0.3, 0.0)),
                              ("ridged", (0.0, 0.5, 0.4))):
                gidx, GG = oracle.group_index(None if kind == "lasso" else groups, p)
                ds.set_groups(None if kind == "lasso" else groups, None if kind == "lasso" else G)
                res = ds.solve_path([pen], tol=1e-12, flags=_engine.FLAG_ON_CHIP, want_group_norms=True)
                assert res.converged and res.mode[0] == 2 and res.grad_launches == 1
                ref, _ = oracle.fista(X, y, pen[0], pen[1], pen[2], gidx, GG, tol=1e-14, max_iter=500000)
                # (p > n: compare objectives -- the minimiser need not be unique)
                f = lambda b: oracle.objective(X, y, b, pen[0], pen[1], pen[2], gidx, GG)  # noqa: E731
                assert f(res.betas[0]) <= f(ref) * (1 + 1e-12) + 1e-14
                if n > p:
                    np.testing.assert_allclose(res.betas[0], ref, rtol=0, atol=1e-9 * np.max(np.abs(ref)))
                np.testing.assert_allclose(res.group_norms[0], np.sqrt(np.bincount(gidx, weights=res.betas[0] ** 2, minlength=GG)),
                                           rtol=1e-12, atol=1e-15)
                # the reported residual is the minimal-norm subgradient at the reported point
                g, _ = ds.gradient(res.betas[0])
                kkt = oracle.kkt_residual(g, res.betas[0], pen[0] * np.ones(p), pen[1] * np.ones(GG), pen[2] * np.ones(GG), gidx, GG)
                assert abs(res.kkt[0] - kkt) <= 1e-8 * max(1.0, np.max(np.abs(g))) and kkt < 1e-8 * np.max(np.abs(g))


def test_paths_folds_and_warm_starts_in_one_launch(eng):
    """Sixteen lanes -- CV folds as row masks with their own 1/n -- each a warm-started 12-point path: one launch, the
    same coefficients as the general path lane by lane; a warm start and COLD_START change nothing but the sweeps."""
    rng = np.random.default_rng(1)
    n, p = 120, 60
    X = rng.standard_normal((n, p)) @ (np.eye(p) + 0.4 * rng.standard_normal((p, p)) / np.sqrt(p))
    y = X[:, :6] @ rng.uniform(1, 3, 6) + rng.standard_normal(n)
    amax = np.max(np.abs(X.T @ y)) / n
    pts = [(a, 0.0, 0.0) for a in np.geomspace(amax, 0.01 * amax, 12)]
    fold = rng.integers(0, 16, n)
    specs = [dict(points=pts, row_weight=(fold != f).astype(float), n_eff=int(np.sum(fold != f))) for f in range(16)]
    with eng.dataset(X, y) as ds:
        assert ds.max_lanes(_engine.FLAG_ON_CHIP) == 16 and ds.max_lanes() < 16
        chip = ds.solve_lanes(specs, tol=1e-11, flags=_engine.FLAG_ON_CHIP)
        for spec, r in zip(specs, chip):
            ref = ds.solve_lanes([spec], tol=1e-11)[0]
            assert r.converged and np.all(r.mode == 2) and ref.converged and np.all(ref.mode != 2)
            np.testing.assert_allclose(r.betas, ref.betas, rtol=0, atol=1e-8 * np.max(np.abs(ref.betas)))
        cold = ds.solve_lanes(specs[:3], tol=1e-11, flags=_engine.FLAG_ON_CHIP | _engine.FLAG_COLD_START)
        warm = ds.solve_lanes([dict(s, beta0=chip[i].betas[3]) for i, s in enumerate(specs[:3])], tol=1e-11, flags=_engine.FLAG_ON_CHIP)
        for i in range(3):
            np.testing.assert_allclose(cold[i].betas, chip[i].betas, rtol=0, atol=1e-8 * np.max(np.abs(chip[i].betas)))
            np.testing.assert_allclose(warm[i].betas, chip[i].betas, rtol=0, atol=1e-8 * np.max(np.abs(chip[i].betas)))
            assert np.sum(cold[i].n_iter) > np.sum(chip[i].n_iter)
        # the shared-path entry point (ranges of one path as lanes) and the flags that name another iteration
        one = ds.solve_path(pts, tol=1e-11, flags=_engine.FLAG_ON_CHIP, lanes=4)
        ref = ds.solve_path(pts, tol=1e-11)
        assert np.all(one.mode == 2)
        np.testing.assert_allclose(one.betas, ref.betas, rtol=0, atol=1e-8 * np.max(np.abs(ref.betas)))
        assert np.all(ds.solve_path(pts, tol=1e-9, flags=_engine.FLAG_ON_CHIP | _engine.FLAG_FISTA_ONLY).mode == 0)


def test_what_does_not_fit_or_does_not_settle_takes_the_general_path(eng):
    rng = np.random.default_rng(2)
    X = rng.standard_normal((40, 200))  # p > 128: no Gram matrix in the LDS
    y = rng.standard_normal(40)
    with eng.dataset(X, y) as ds:
        r = ds.solve_path([(0.3, 0.0, 0.0)], flags=_engine.FLAG_ON_CHIP)
        assert r.converged and r.mode[0] != 2
    # nearly collinear columns: the coordinate iteration crawls; with two sweeps allowed it must hand over, and the
    # answer is the general path's
    n, p = 60, 12
    base = rng.standard_normal((n, 3))
    X = base @ rng.standard_normal((3, p)) + 1e-4 * rng.standard_normal((n, p))
    y = X @ rng.standard_normal(p) + 0.1 * rng.standard_normal(n)
    with eng.dataset(X, y) as ds:
        ref = ds.solve_path([(1e-3, 0.0, 0.0)], tol=1e-9, max_iter=200000)
        r = ds.solve_path([(1e-3, 0.0, 0.0)], tol=1e-9, max_iter=200000, flags=_engine.FLAG_ON_CHIP)
        assert r.converged and ref.converged
        f = lambda b: 0.5 * np.mean((X @ b - y) ** 2) + 1e-3 * np.sum(np.abs(b))  # noqa: E731
        assert f(r.betas[0]) <= f(ref.betas[0]) * (1 + 1e-9)
    with pytest.raises(_engine.NonFiniteError):
        bad = X.copy()
        bad[3, 2] = np.inf
        with eng.dataset(bad, y) as ds:
            ds.solve_path([(0.1, 0.0, 0.0)], flags=_engine.FLAG_ON_CHIP)


def test_randomised_cross_check_against_the_general_path():
    """tools/on_chip_fuzz.py: 150 random problems of up to 128 features (penalty kinds, groups, weighted l1, ridge, fold
    masks, paths, warm starts, p > n, correlated columns), the on-chip call against the same cells on the general path."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "on_chip_fuzz.py"), "150", "3"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "flagged 0" in out.stdout.splitlines()[-1]
