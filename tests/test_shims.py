"""The thin layers beside the fit path: dataset generator, OrdinaryLeastSquares, the composite
estimator with a search inside.  Mirrors /root/reference/tests/test_dataset.py, test_ols.py and
test_stepwise.py (MIQP steps replaced by in-scope estimators)."""

import warnings

import numpy as np
import numpy.testing as npt
import pytest
from sklearn.base import clone

from _oracle_backend import OracleBackend
from sparselm_amd import _backend
from sparselm_amd.dataset import make_group_regression
from sparselm_amd.model import GroupLasso, Lasso, OrdinaryLeastSquares
from sparselm_amd.model_selection import GridSearchCV
from sparselm_amd.stepwise import StepwiseEstimator

TIGHT = {"tol": 1e-12, "max_iter": 200000}


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    if request.param == "oracle":
        with _backend.use_backend(OracleBackend()):
            yield "oracle"
    else:
        yield "hip"


# ---- dataset ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_informative_groups", [5, 20])
@pytest.mark.parametrize("n_features_per_group", [5, 4 * list(range(2, 7))])
@pytest.mark.parametrize("frac_informative_in_group", [1.0, 0.5])
@pytest.mark.parametrize("shuffle", [True, False])
@pytest.mark.parametrize("coef", [True, False])
def test_make_group_regression(n_informative_groups, n_features_per_group, frac_informative_in_group, shuffle, coef):
    model = make_group_regression(
        n_informative_groups=n_informative_groups,
        n_features_per_group=n_features_per_group,
        frac_informative_in_group=frac_informative_in_group,
        shuffle=shuffle,
        coef=coef,
        random_state=0,
    )
    assert len(model) == (4 if coef else 3)
    X, y, groups = model[:3]
    sizes = n_features_per_group if isinstance(n_features_per_group, list) else [n_features_per_group] * 20
    n_features = sum(sizes)
    assert X.shape == (100, n_features) and y.shape == (100,) and groups.shape == (n_features,)
    assert len(np.unique(groups)) == 20
    if coef:
        coefs = model[3]
        n_informative = sum(round(frac_informative_in_group * sizes[i]) for i in range(n_informative_groups))
        assert coefs.shape == (n_features,)
        assert int(np.sum(coefs > 0)) == n_informative
        npt.assert_array_almost_equal(X @ coefs, y)
        # informative features sit in n_informative_groups groups
        assert len(np.unique(groups[coefs > 0])) == n_informative_groups
    if shuffle:
        assert np.sum(np.diff(groups) == 0) < n_features - 20  # labels are not lumped together
    else:
        assert np.all(np.diff(groups) >= 0)


def test_make_group_regression_warns_and_validates():
    with pytest.warns(UserWarning):
        make_group_regression(frac_informative_in_group=1 / 100)
    with pytest.raises(ValueError):
        make_group_regression(n_groups=4, n_features_per_group=[2, 3])


# ---- ordinary least squares -------------------------------------------------------------------------
def test_linear_regression(backend):
    reg = OrdinaryLeastSquares(solver_options=TIGHT).fit([[1], [2]], [1, 2])
    npt.assert_array_almost_equal(reg.coef_, [1])
    npt.assert_array_almost_equal(reg.intercept_, [0])
    npt.assert_array_almost_equal(reg.predict([[1], [2]]), [1, 2])
    reg = OrdinaryLeastSquares(solver_options=TIGHT).fit([[1]], [0])  # degenerate input
    npt.assert_array_almost_equal(reg.coef_, [0])
    npt.assert_array_almost_equal(reg.intercept_, [0])
    npt.assert_array_almost_equal(reg.predict([[1]]), [0])


def test_fit_intercept_shapes(backend):
    X2 = np.array([[0.38349978, 0.61650022], [0.58853682, 0.41146318]])
    X3 = np.array([[0.27677969, 0.70693172, 0.01628859], [0.08385139, 0.20692515, 0.70922346]])
    y = np.array([1, 1])
    a2, b2 = OrdinaryLeastSquares(fit_intercept=False).fit(X2, y), OrdinaryLeastSquares().fit(X2, y)
    a3, b3 = OrdinaryLeastSquares(fit_intercept=False).fit(X3, y), OrdinaryLeastSquares().fit(X3, y)
    assert a2.coef_.shape == b2.coef_.shape and a3.coef_.shape == b3.coef_.shape
    assert a2.coef_.ndim == a3.coef_.ndim == 1


# ---- composite estimator ----------------------------------------------------------------------------
def test_make_composite():
    lasso1 = Lasso(fit_intercept=True, alpha=1.0)
    lasso2 = Lasso(fit_intercept=False, alpha=2.0)
    gl = GroupLasso(groups=[0, 0, 1, 2], alpha=0.1)
    scopes = [[0, 1, 8], [2, 3], [4, 5, 6, 7]]
    est = StepwiseEstimator([("lasso1", lasso1), ("lasso2", lasso2), ("gl", gl)], scopes)
    assert est.steps[0][1].fit_intercept and not est.steps[1][1].fit_intercept and not est.steps[2][1].fit_intercept
    params = est.get_params(deep=True)
    assert params["lasso1"].get_params()["alpha"] == 1.0 and params["lasso1__alpha"] == 1.0
    assert params["lasso2__alpha"] == 2.0 and params["gl__alpha"] == 0.1
    est.set_params(lasso2__alpha=0.5, gl__alpha=0.2)
    params = est.get_params(deep=True)
    assert params["lasso1__alpha"] == 1.0 and params["lasso2__alpha"] == 0.5 and params["gl__alpha"] == 0.2
    cloned = clone(est)
    params = cloned.get_params(deep=True)
    assert params["lasso2"].get_params()["alpha"] == 0.5 and params["gl__alpha"] == 0.2
    # a searcher as a step
    grid = GridSearchCV(lasso2, {"alpha": [0.01, 0.1, 1.0]})
    est = StepwiseEstimator([("lasso1", lasso1), ("lasso2", grid), ("gl", gl)], scopes)
    params = est.get_params(deep=True)
    assert params["lasso1__alpha"] == 1.0 and params["gl__alpha"] == 0.2
    assert "lasso2__alpha" not in params and params["lasso2__estimator__alpha"] == 0.5


def test_toy_composite(backend):
    from sklearn.utils._param_validation import InvalidParameterError

    rng = np.random.default_rng(5)
    lasso1 = Lasso(fit_intercept=True, alpha=1e-6, solver_options=TIGHT)
    lasso2 = Lasso(fit_intercept=False, alpha=1e-6, solver_options=TIGHT)
    grid = GridSearchCV(clone(lasso2), {"alpha": [1e-8, 1e-7, 1e-6]})
    gl = GroupLasso(groups=[0, 0, 1, 2], alpha=1e-9, solver_options=TIGHT)
    scopes = [[0, 1, 8], [2, 3], [4, 5, 6, 7]]
    est = StepwiseEstimator([("lasso1", lasso1), ("lasso2", lasso2), ("gl", gl)], scopes)
    est2 = StepwiseEstimator([("lasso1", clone(lasso1)), ("lasso2", grid), ("gl", clone(gl))], scopes)
    w = rng.normal(scale=2, size=9) * 0.2
    w[0], w[-1] = 10, 0.5
    X = rng.random(size=(20, 9))
    X[:, 0] = 1
    X[:, -1] = -8 * rng.random(size=20)
    y = X @ w + rng.normal(scale=0.01, size=20)
    with pytest.raises(ValueError):  # too many features for the scopes
        est.fit(rng.random(size=(20, 12)), rng.random(size=20))
    with pytest.raises(InvalidParameterError):  # scopes that do not cover the features
        StepwiseEstimator(est.steps, [[0, 1], [3, 4], [5, 6, 7, 8]]).fit(X, y)
    with pytest.raises(InvalidParameterError):  # an intercept beyond the first step
        StepwiseEstimator([("a", lasso1), ("b", Lasso(fit_intercept=True, alpha=1e-6)), ("c", gl)], scopes).fit(X, y)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for e in (est, est2):
            e.fit(X, y)
            assert e.intercept_ == e.steps[0][1].intercept_ and not np.isclose(e.intercept_, 0)
            assert not np.any(np.isnan(e.coef_))
            for (_, sub), scope in zip(e.steps, e.estimator_feature_indices):
                sub_coef = sub.best_estimator_.coef_ if hasattr(sub, "estimator") else sub.coef_
                npt.assert_array_almost_equal(sub_coef, e.coef_[list(scope)])
            # the first step sees the constant column and the big coefficient: it explains most of y
            assert np.mean((e.predict(X) - y) ** 2) < 0.3 * np.var(y)
