"""Certified partial passes (csrc/light_kernels.hpp): the re-verification of a lane after a miss without a pass over X.

A point re-verified this way is accepted on a gradient that is exact on the working set (from its Gram) and on the
borderline columns (from their own rows of the column-major copy) and PROVABLY irrelevant elsewhere (Cauchy-Schwarz with
the column norms): the same proximal-gradient mapping, the same stopping rule.  The tests hold that against the route it
replaces (SLM_NO_LIGHT_PASS=1: every re-verification a pass over X) and against the oracle (oracle.fista, the objective of
/root/reference/src/sparselm/model/_lasso.py:99-121), check the pieces -- column norms, the certificate's inequality on
every column it did not read -- and count the passes saved.
"""

import numpy as np
import pytest

import oracle
from sparselm_amd import _engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return _engine.get_engine(0)


def _noisy_problem(n, p, k, seed, noise=3.0):
    """A path whose END sits at the noise floor: features no earlier gradient can tell enter at its last points."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p))
    beta = np.zeros(p)
    beta[rng.choice(p, k, replace=False)] = 30.0 * rng.uniform(0.1, 1.0, k)
    y = X @ beta + noise * rng.standard_normal(n)
    return X, y


def _path(ds, X, y, K=50, floor=1e-3, lanes=18, **kw):
    n = len(y)
    amax = float(np.max(np.abs(X.T @ y)) / n)
    alphas = np.geomspace(amax, floor * amax, K)
    return alphas, ds.solve_path([(a, 0.0, 0.0) for a in alphas], lanes=lanes, flags=_engine.FLAG_WORKING_SET | _engine.FLAG_FRESH_L, **kw)


def test_column_norms_and_light_passes_against_passes_over_x(eng, monkeypatch):
    used = saved = 0
    for seed in range(6):
        X, y = _noisy_problem(12_000, 1500, 25, seed)
        n, p = X.shape
        with eng.dataset(X, y) as ds:
            alphas, light = _path(ds, X, y)
            monkeypatch.setenv("SLM_NO_LIGHT_PASS", "1")
            _, full = _path(ds, X, y)
            monkeypatch.delenv("SLM_NO_LIGHT_PASS")
            _, again = _path(ds, X, y)
        assert light.converged and full.converged
        assert full.light_passes == 0
        assert np.array_equal(light.betas, again.betas) and light.light_passes == again.light_passes  # run to run
        # the same minimisers: both routes accept a point under the same rule
        top = np.max(np.abs(full.betas), axis=1)
        for k in range(len(alphas)):
            if top[k] > 0:
                assert np.max(np.abs(light.betas[k] - full.betas[k])) <= 2e-7 * top[k], (seed, k)
        # a light pass stands in for a pass over X: never more reads of X.  (The two routes need not take the same NUMBER of
        # verifications: behind a light pass the scores of the next selection see the base point's gradient off the
        # borderline set, so the sets grow differently -- the minimisers are the same, above.)
        assert light.grad_launches <= full.grad_launches
        used += light.light_passes
        saved += full.grad_launches - light.grad_launches
        if light.light_passes:
            assert light.light_columns <= 1024 * light.light_passes
        # ... and the oracle on three points of the path, the last among them
        gidx, G = oracle.group_index(None, p)
        b = None
        for k in (10, 30, len(alphas) - 1):
            b, _ = oracle.fista(X, y, alphas[k], 0.0, 0.0, gidx, G, beta0=b, tol=1e-13)
            assert np.max(np.abs(light.betas[k] - b)) <= 1e-6 * np.max(np.abs(b)), (seed, k)
    assert used > 0 and saved > 0, (used, saved)  # (six noisy paths: some end with a miss)


def test_every_unread_column_satisfies_the_certificate(eng):
    """After a path that used light passes: at its last point the optimality conditions hold on EVERY column with a gradient
    numpy computes from X -- the columns the certificate skipped included."""
    hits = 0
    for seed in range(6, 12):
        X, y = _noisy_problem(10_000, 1200, 20, seed)
        n, p = X.shape
        with eng.dataset(X, y) as ds:
            alphas, res = _path(ds, X, y, K=36)
        assert res.converged
        if not res.light_passes:
            continue
        hits += 1
        for k in (len(alphas) - 1, len(alphas) - 2):
            beta = res.betas[k]
            g = X.T @ (X @ beta - y) / n
            on = beta != 0
            assert np.max(np.abs(g[~on])) <= alphas[k] * (1 + 1e-7), (seed, k)
            assert np.max(np.abs(g[on] + alphas[k] * np.sign(beta[on]))) <= 1e-6 * alphas[0], (seed, k)
    assert hits > 0


def test_weighted_l1_paths_take_light_passes_with_their_own_thresholds(eng, monkeypatch):
    """Per-feature weights (the penalty of an AdaptiveLasso round, model/_adaptive_lasso.py:158-232: a_j = alpha * w_j): the
    certificate compares with a_j, feature by feature.  Against the route without light passes and the oracle."""
    used = 0
    for seed in range(20, 26):
        X, y = _noisy_problem(12_000, 1500, 25, seed)
        n, p = X.shape
        w = np.random.default_rng(seed).uniform(0.5, 2.0, p)
        amax = float(np.max(np.abs(X.T @ y) / w) / n)
        alphas = np.geomspace(amax, 1e-3 * amax, 50)
        pts = [(a, 0.0, 0.0) for a in alphas]
        flags = _engine.FLAG_WORKING_SET | _engine.FLAG_FRESH_L
        with eng.dataset(X, y) as ds:
            light = ds.solve_path(pts, a=w, lanes=18, flags=flags)
            monkeypatch.setenv("SLM_NO_LIGHT_PASS", "1")
            full = ds.solve_path(pts, a=w, lanes=18, flags=flags)
            monkeypatch.delenv("SLM_NO_LIGHT_PASS")
        assert light.converged and full.converged and full.light_passes == 0
        assert light.grad_launches <= full.grad_launches
        used += light.light_passes
        top = np.max(np.abs(full.betas), axis=1)
        for k in range(len(alphas)):
            if top[k] > 0:
                assert np.max(np.abs(light.betas[k] - full.betas[k])) <= 2e-7 * top[k], (seed, k)
        gidx, G = oracle.group_index(None, p)
        b = None
        for k in (15, 35, 49):
            b, _ = oracle.fista(X, y, alphas[k] * w, 0.0, 0.0, gidx, G, beta0=b, tol=1e-13)
            assert np.max(np.abs(light.betas[k] - b)) <= 1e-6 * np.max(np.abs(b)), (seed, k)
    assert used > 0


def test_group_paths_take_light_passes_too(eng, monkeypatch):
    """Group and sparse-group penalties: a group outside the working set is certified as a whole -- the 2-norm of the largest
    soft-thresholded gradient its coordinates can have stays below its weight -- or all its members are read.  Against the
    route without light passes and against the oracle (model/_lasso.py:230-275, 616-639)."""
    used = 0
    # (small problems run the fused gradient kernels, not the split pass: the attempt stands in front of either)
    for seed, l1 in ((0, 0.0), (2, 0.3), (4, 0.5), (6, 0.7)):
        rng = np.random.default_rng(100 + seed)
        n, p, gs = 9000, 1200, 10
        G = p // gs
        X = rng.standard_normal((n, p))
        groups = rng.permutation(np.repeat(np.arange(G), gs)).astype(np.int32)
        beta = np.zeros(p)
        for g in rng.choice(G, 8, replace=False):
            beta[groups == g] = rng.uniform(1.0, 4.0, gs) * rng.choice([-1.0, 1.0], gs)
        y = X @ beta + 4.0 * rng.standard_normal(n)
        g0 = X.T @ y / n
        bmax = float(np.max(np.sqrt(np.bincount(groups, weights=g0 * g0, minlength=G))))
        scales = np.geomspace(bmax, 1e-2 * bmax, 30)
        pts = [(l1 * a, (1.0 - l1) * a, 0.0) for a in scales]
        with eng.dataset(X, y) as ds:
            ds.set_groups(groups, G)
            # (three lanes, ten points each: a lane that misses late in its range re-verifies past the expected end)
            light = ds.solve_path(pts, lanes=3, flags=_engine.FLAG_WORKING_SET | _engine.FLAG_FRESH_L)
            monkeypatch.setenv("SLM_NO_LIGHT_PASS", "1")
            full = ds.solve_path(pts, lanes=3, flags=_engine.FLAG_WORKING_SET | _engine.FLAG_FRESH_L)
            monkeypatch.delenv("SLM_NO_LIGHT_PASS")
        assert light.converged and full.converged and full.light_passes == 0
        assert light.grad_launches <= full.grad_launches
        used += light.light_passes
        gidx, Gn = oracle.group_index(groups, p)
        b = None
        for k in (8, 20, 29):
            b, _ = oracle.fista(X, y, pts[k][0], pts[k][1], 0.0, gidx, Gn, beta0=b, tol=1e-13)
            top = float(np.max(np.abs(b)))
            if top > 0:
                assert np.max(np.abs(light.betas[k] - b)) <= 1e-6 * top, (seed, k)
                assert np.max(np.abs(full.betas[k] - b)) <= 1e-6 * top, (seed, k)
    assert used > 0
