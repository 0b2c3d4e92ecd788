"""Covariance passes (SLM_FLAG_COVARIANCE, csrc/cov_kernels.hpp): gradients from the Gram of a row set instead of a
read of X -- for the many solves of a CV grid that share a fold (the reference's loop over candidates x folds,
/root/reference/src/sparselm/model_selection.py:304-323).  Same iteration, same results."""

import numpy as np
import pytest

from sparselm_amd import _engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    return _engine.get_engine(0)


def _problem(n, p, seed=0):
    rng = np.random.default_rng(seed)
    G = p // 10
    groups = rng.permutation(np.repeat(np.arange(G), 10))
    coef = np.zeros(p)
    for g in rng.choice(G, 6, replace=False):
        coef[groups == g] = 10.0 * rng.uniform(size=10)
    X = rng.standard_normal((n, p))
    y = X @ coef + 5.0 * rng.standard_normal(n)
    return X, y, groups, G, rng


def test_sixteen_lane_calls_from_the_grams_of_their_folds(eng):
    """Sixteen lanes -- five CV folds as row masks with their own 1/n, sparse-group paths of twelve points -- over X and
    from the folds' Grams: the same passes, the same coefficients to rounding; a Gram is found again by the content of
    its row weights (another array with the same values), general weights and the dataset's own rows work too, new
    targets drop what was built."""
    n, p = 6000, 400
    X, y, groups, G, rng = _problem(n, p)
    folds = rng.permutation(n) % 5
    masks = [(folds != f).astype(float) for f in range(5)]
    with eng.dataset(X, y) as ds:
        ds.set_groups(groups, G)
        g0, _ = ds.gradient(None)
        amax = float(np.max(np.abs(g0)))
        al = np.geomspace(amax, 1e-2 * amax, 12)
        pts = np.c_[0.5 * al, 0.5 * al, 0 * al]
        specs = [dict(points=pts * (1.0 + 0.1 * (l // 5)), row_weight=masks[l % 5], n_eff=int(masks[l % 5].sum())) for l in range(16)]
        base = _engine.FLAG_WORKING_SET
        ref = ds.solve_lanes(specs, tol=1e-10, flags=base)
        # the flag without Grams changes nothing
        same = ds.solve_lanes(specs, tol=1e-10, flags=base | _engine.FLAG_COVARIANCE)
        assert all(np.array_equal(a.betas, b.betas) for a, b in zip(ref, same))
        for m in masks:
            ds.covariance(m, int(m.sum()))
        assert ds.covariance_count() == 5
        ds.covariance(masks[3].copy(), int(masks[3].sum()))  # same content, another array: found, not built again
        assert ds.covariance_count() == 5
        cov = ds.solve_lanes(specs, tol=1e-10, flags=base | _engine.FLAG_COVARIANCE)
        for a, b in zip(ref, cov):
            assert a.converged and b.converged and a.grad_launches == b.grad_launches
            np.testing.assert_allclose(b.betas, a.betas, rtol=0, atol=1e-9 * np.max(np.abs(a.betas)))
            np.testing.assert_allclose(b.loss, a.loss, rtol=1e-9)
        # a call with a row set that has no Gram runs over X (on the split pass, which the flag asks for: equal to rounding)
        other = (rng.permutation(n) % 3 != 0).astype(float)
        mixed = specs[:3] + [dict(points=pts, row_weight=other, n_eff=int(other.sum()))]
        a = ds.solve_lanes(mixed, tol=1e-10, flags=base)
        b = ds.solve_lanes(mixed, tol=1e-10, flags=base | _engine.FLAG_COVARIANCE)
        for u, v in zip(a, b):
            np.testing.assert_allclose(v.betas, u.betas, rtol=0, atol=1e-9 * np.max(np.abs(u.betas)))
        # general weights, and all rows with the dataset's 1/n
        w = rng.uniform(0.5, 1.5, n)
        ds.covariance(w, 0)
        ds.covariance(None, 0)
        assert ds.covariance_count() == 7
        for rw in (w, None):
            three = [dict(points=pts, row_weight=rw), dict(points=1.3 * pts, row_weight=rw), dict(points=0.8 * pts, row_weight=rw)]
            a = ds.solve_lanes(three, tol=1e-10, flags=base)
            b = ds.solve_lanes(three, tol=1e-10, flags=base | _engine.FLAG_COVARIANCE)
            for u, v in zip(a, b):
                assert u.converged and v.converged
                np.testing.assert_allclose(v.betas, u.betas, rtol=0, atol=1e-9 * np.max(np.abs(u.betas)))
        ds.set_targets(2.0 * y)
        assert ds.covariance_count() == 0


def test_grid_search_from_the_folds_grams(eng):
    """GridSearchCV(SparseGroupLasso) with solver_options covariance=True / False: the same table of scores and the same
    choice.  "auto" does not ask for Grams on a grid this small -- and, since round 5, does not take the ones an earlier search
    left on the cached dataset either (the searches of one (X, y) share a device dataset through the cache, and Grams of some
    other split on it used to switch the cost model off: every fold of the new split was then built mask by mask; only the
    lines of ONE LineSearchCV -- a leased dataset, same masks by construction -- take over each other's Grams)."""
    from sparselm_amd.model import SparseGroupLasso
    from sparselm_amd.model_selection import GridSearchCV

    n, p = 3000, 200
    X, y, groups, G, _ = _problem(n, p, seed=3)
    grid = {"alpha": np.geomspace(2.0, 0.05, 8), "l1_ratio": [0.2, 0.5, 0.8]}
    out = {}
    for cov in (False, "auto", True, "auto again"):
        opts = {"tol": 1e-10, "covariance": "auto" if isinstance(cov, str) else cov, "on_chip": False}
        gs = GridSearchCV(SparseGroupLasso(groups=groups, solver_options=opts), grid, cv=4).fit(X, y)
        out[cov] = gs
    np.testing.assert_allclose(out[True].cv_results_["mean_test_score"], out[False].cv_results_["mean_test_score"], rtol=1e-8)
    np.testing.assert_array_equal(out["auto"].cv_results_["mean_test_score"], out[False].cv_results_["mean_test_score"])
    # (the Grams of the third search are on the cached dataset: the fourth decides by its own cost model -- over X, the second's table)
    np.testing.assert_array_equal(out["auto again"].cv_results_["mean_test_score"], out["auto"].cv_results_["mean_test_score"])
    assert out[True].best_params_ == out[False].best_params_
    np.testing.assert_allclose(out[True].best_estimator_.coef_, out[False].best_estimator_.coef_, rtol=0,
                               atol=1e-8 * np.max(np.abs(out[False].best_estimator_.coef_)))


def test_adaptive_grid_search_from_the_folds_grams(eng):
    """GridSearchCV(AdaptiveLasso): every cell is a loop of re-weighted solves on its fold -- with covariance=True all of
    them read the fold's Gram; same scores, same number of re-weighting rounds in the refit."""
    import warnings

    from sparselm_amd.model import AdaptiveLasso
    from sparselm_amd.model_selection import GridSearchCV

    n, p = 2500, 160
    X, y, _, _, _ = _problem(n, p, seed=5)
    grid = {"alpha": np.geomspace(1.0, 0.02, 6)}
    out = {}
    for cov in (False, True):
        opts = {"tol": 1e-10, "covariance": cov, "on_chip": False}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out[cov] = GridSearchCV(AdaptiveLasso(fit_intercept=True, solver_options=opts), grid, cv=3).fit(X, y)
    np.testing.assert_allclose(out[True].cv_results_["mean_test_score"], out[False].cv_results_["mean_test_score"], rtol=1e-7)
    assert out[True].best_params_ == out[False].best_params_
    assert out[True].best_estimator_.n_iter_ == out[False].best_estimator_.n_iter_


def test_copies_on_further_engines_share_the_grams(eng):
    """slm_dataset_clone hands the Grams built so far to the copy (same device, read-only, freed with the last holder): the
    streams of a grid search read them too.  The copy outlives the original."""
    n, p = 4000, 240
    X, y, groups, G, rng = _problem(n, p, seed=9)
    mask = (rng.permutation(n) % 4 != 0).astype(float)
    ds = eng.dataset(X, y)
    ds.set_groups(groups, G)
    g0, _ = ds.gradient(None)
    al = np.geomspace(float(np.max(np.abs(g0))), 0.01 * float(np.max(np.abs(g0))), 8)
    specs = [dict(points=np.c_[0.5 * al, 0.5 * al, 0 * al] * s, row_weight=mask, n_eff=int(mask.sum())) for s in (1.0, 1.2, 0.9)]
    ref = ds.solve_lanes(specs, tol=1e-10, flags=_engine.FLAG_WORKING_SET)
    ds.covariance(mask, int(mask.sum()))
    other = _engine.Engine(0)
    try:
        copy = ds.clone(other)
        copy.set_groups(groups, G)
        assert copy.covariance_count() == 1
        ds.close()  # the copy keeps the blocks alive
        out = copy.solve_lanes(specs, tol=1e-10, flags=_engine.FLAG_WORKING_SET | _engine.FLAG_COVARIANCE)
        for a, b in zip(ref, out):
            assert b.converged
            np.testing.assert_allclose(b.betas, a.betas, rtol=0, atol=1e-9 * np.max(np.abs(a.betas)))
        copy.close()
    finally:
        other.close()


def test_the_folds_of_a_split_at_once(eng):
    """slm_dataset_covariance_folds: the test rows of a K-fold split partition the rows, so the Gram of all rows is the sum of
    their Grams -- same entries as mask by mask (found again by either call), same solves; masks that are no partition fall
    back to the mask-by-mask build."""
    n, p = 5000, 300
    X, y, groups, G, rng = _problem(n, p, seed=4)
    folds = rng.permutation(n) % 4
    masks = [(folds != f).astype(float) for f in range(4)]
    nes = [int(m.sum()) for m in masks]
    with eng.dataset(X, y) as a, eng.dataset(X, y) as b:
        for ds in (a, b):
            ds.set_groups(groups, G)
        a.covariance_folds(masks, nes)
        assert a.covariance_count() == 4
        a.covariance(masks[2], nes[2])  # found, not built again
        assert a.covariance_count() == 4
        for m, ne in zip(masks, nes):
            b.covariance(m, ne)
        g0, _ = a.gradient(None)
        al = np.geomspace(float(np.max(np.abs(g0))), 0.02 * float(np.max(np.abs(g0))), 8)
        specs = [dict(points=np.c_[0.4 * al, 0.6 * al, 0 * al] * (1 + 0.1 * (l // 4)), row_weight=masks[l % 4], n_eff=nes[l % 4]) for l in range(12)]
        flags = _engine.FLAG_WORKING_SET | _engine.FLAG_COVARIANCE
        ra, rb = a.solve_lanes(specs, tol=1e-10, flags=flags), b.solve_lanes(specs, tol=1e-10, flags=flags)
        ref = a.solve_lanes(specs, tol=1e-10, flags=_engine.FLAG_WORKING_SET)
        for u, v, w in zip(ra, rb, ref):
            assert u.converged and v.converged
            scale = np.max(np.abs(w.betas))
            assert np.max(np.abs(u.betas - v.betas)) < 1e-9 * scale and np.max(np.abs(u.betas - w.betas)) < 1e-9 * scale
    # no partition: overlapping test rows, and a weight that is not 0/1
    with eng.dataset(X, y) as ds:
        odd = [masks[0], (rng.permutation(n) % 3 != 0).astype(float), rng.uniform(0.5, 1.5, n)]
        ds.covariance_folds(odd, [int(odd[0].sum()), int(odd[1].sum()), n])
        assert ds.covariance_count() == 3
        one = [dict(points=[(0.1, 0.0, 0.0)], row_weight=odd[2], n_eff=n)]
        u = ds.solve_lanes(one, tol=1e-10, flags=_engine.FLAG_WORKING_SET)[0]
        v = ds.solve_lanes(one, tol=1e-10, flags=_engine.FLAG_WORKING_SET | _engine.FLAG_COVARIANCE)[0]
        np.testing.assert_allclose(v.betas, u.betas, rtol=0, atol=1e-9 * np.max(np.abs(u.betas)))


def test_a_single_fit_from_the_gram_when_asked(eng):
    """solver_options={"covariance": True} on an estimator: the fit's passes read the Gram of the (cached) device dataset;
    same coefficients as the default fit."""
    from sparselm_amd.model import AdaptiveLasso, Lasso

    n, p = 3000, 250
    X, y, _, _, _ = _problem(n, p, seed=8)
    for cls, kw in ((Lasso, dict(alpha=0.3)), (AdaptiveLasso, dict(alpha=0.3, max_iter=4))):
        a = cls(fit_intercept=True, solver_options={"tol": 1e-10, "on_chip": False}, **kw).fit(X, y)
        b = cls(fit_intercept=True, solver_options={"tol": 1e-10, "on_chip": False, "covariance": True}, **kw).fit(X, y)
        np.testing.assert_allclose(b.coef_, a.coef_, rtol=0, atol=1e-8 * np.max(np.abs(a.coef_)))
        np.testing.assert_allclose(b.intercept_, a.intercept_, rtol=1e-8)
