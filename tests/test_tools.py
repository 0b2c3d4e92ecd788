"""Box-constrained refits and the r2 -> CV error helper (mirrors /root/reference/tests/test_tools.py).

Runs with the CPU oracle behind the surface here and through the HIP engine under ``-m gpu``.
"""

import warnings
from functools import partial

import numpy as np
import numpy.testing as npt
import pytest

from _oracle_backend import OracleBackend
from sparselm_amd import _backend
from sparselm_amd.model import Lasso, OrdinaryLeastSquares
from sparselm_amd.tools import constrain_coefficients, r2_score_to_cv_error


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    if request.param == "oracle":
        with _backend.use_backend(OracleBackend()):
            yield "oracle"
    else:
        yield "hip"


def _fit(X, y, reg):
    reg.fit(X, y)
    return reg.coef_


def _check(inds, low, high, X, y, reg, n_coefs):
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        coefs = constrain_coefficients(inds, high=high, low=low)(partial(_fit, reg=reg))(X, y)
    assert coefs.shape == (n_coefs,)
    if any(issubclass(w.category, RuntimeWarning) for w in caught):
        with pytest.warns(RuntimeWarning):
            constrain_coefficients(inds, high=high, low=low)(partial(_fit, reg=reg))(X, y)
    else:
        lo = np.full(len(inds), -np.inf) if low is None else np.broadcast_to(low, (len(inds),))
        hi = np.full(len(inds), np.inf) if high is None else np.broadcast_to(high, (len(inds),))
        assert np.all(coefs[inds] >= lo - 1e-12) and np.all(coefs[inds] <= hi + 1e-12)

    @constrain_coefficients(inds, high, low)
    def decorated(X, y, reg):
        reg.fit(X, y)
        return reg.coef_

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        npt.assert_almost_equal(decorated(X, y, reg=reg), coefs)
    return coefs


@pytest.mark.parametrize("trial", range(5))
def test_constrain_coefficients(trial, backend):
    rng = np.random.default_rng(100 + trial)
    n_samples, n_features = 10, 8
    X = rng.normal(size=(n_samples, n_features))
    y = rng.normal(size=n_samples)
    reg = OrdinaryLeastSquares(fit_intercept=True, solver_options={"tol": 1e-12, "max_iter": 200000})
    inds = rng.choice(n_features, size=3, replace=False)

    _check(inds, 0, 2, X, y, reg, n_features)  # uniform bounds
    low = rng.random(size=3) - 0.5
    high = rng.random(size=3) + low
    _check(inds, low, high, X, y, reg, n_features)  # per-index bounds
    _check(inds, None, high, X, y, reg, n_features)  # only an upper bound
    _check(inds, low, None, X, y, reg, n_features)  # only a lower bound


def test_constrained_fit_pins_violators_and_refits(backend):
    # the refit solves the reduced problem exactly: pinned columns move to the right-hand side
    rng = np.random.default_rng(7)
    X = rng.normal(size=(60, 6))
    beta = np.array([3.0, -2.0, 0.5, 0.0, 1.0, -1.0])
    y = X @ beta
    reg = Lasso(alpha=1e-10, fit_intercept=False, solver_options={"tol": 1e-13, "max_iter": 200000})
    coefs = constrain_coefficients([0, 1], high=1.0, low=-1.0)(partial(_fit, reg=reg))(X, y)
    assert coefs[0] == 1.0 and coefs[1] == -1.0
    free = [2, 3, 4, 5]
    expected = np.linalg.lstsq(X[:, free], y - X[:, 0] * 1.0 - X[:, 1] * -1.0, rcond=None)[0]
    npt.assert_allclose(coefs[free], expected, atol=1e-6)


def test_constrain_coefficients_leaves_inputs_untouched_and_validates_bounds(backend):
    rng = np.random.default_rng(3)
    X = rng.normal(size=(20, 4))
    y = X @ np.array([5.0, 0.0, -5.0, 1.0])
    X0, y0 = X.copy(), y.copy()
    reg = OrdinaryLeastSquares(fit_intercept=False)
    constrain_coefficients([0, 2], 1, -1)(partial(_fit, reg=reg))(X, y)
    npt.assert_array_equal(X, X0)
    npt.assert_array_equal(y, y0)
    with pytest.raises(ValueError):
        constrain_coefficients([0, 2], high=[1.0, 2.0, 3.0])


def test_r2_score_to_cv_error():
    rng = np.random.default_rng(0)
    y = rng.normal(size=50)
    y_pred = y + 0.1 * rng.normal(size=50)
    w = rng.random(50) + 0.1
    # with score = r2 of the same predictions the helper returns sqrt(SS_res/SS_tot * MSE_w)
    mse = np.sum(w * (y - y_pred) ** 2) / np.sum(w)
    assert r2_score_to_cv_error(0.0, y, y_pred, w) == pytest.approx(np.sqrt(mse))
    assert r2_score_to_cv_error(0.75, y, y_pred) == pytest.approx(0.5 * np.sqrt(np.mean((y - y_pred) ** 2)))
    assert r2_score_to_cv_error(1.0, y, y_pred) == 0.0
    with pytest.raises(ValueError):
        r2_score_to_cv_error(0.5, y, y_pred, w[:-1])
    with pytest.raises(ValueError):
        r2_score_to_cv_error(0.5, y, y_pred, -w)
    with pytest.raises(ValueError):
        r2_score_to_cv_error(0.5, y, y_pred, np.zeros(50))
