"""The reference's own helper code running on top of sparselm_amd estimators.

Runs only where /root/reference is mounted (the build container), on CPU with the oracle behind the
surface; nothing from the reference is copied into this repo.
"""

import os
import sys

import numpy as np
import pytest

REF_SRC = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="reference tree not mounted")


@pytest.fixture()
def ref_tools():
    sys.path.insert(0, REF_SRC)
    try:
        import sparselm.tools as tools  # numpy-only module of the reference

        yield tools
    finally:
        sys.path.remove(REF_SRC)
        for name in [m for m in sys.modules if m == "sparselm" or m.startswith("sparselm.")]:
            del sys.modules[name]


def test_reference_constrain_coefficients_wraps_our_fit(ref_tools):
    # reference src/sparselm/tools.py:14-97 used the way its docstring shows, around our estimator
    from _oracle_backend import OracleBackend
    from sparselm_amd import _backend
    from sparselm_amd.model import Lasso

    rng = np.random.default_rng(0)
    X = rng.standard_normal((60, 6))
    y = X @ np.array([3.0, -2.0, 0.5, 0.0, 0.0, 1.0]) + 0.01 * rng.standard_normal(60)

    def fit_method(X, y):
        return Lasso(alpha=1e-3, solver_options={"tol": 1e-12}).fit(X, y).coef_

    with _backend.use_backend(OracleBackend()):
        free = fit_method(X, y)
        assert free[0] > 2.5
        coefs = ref_tools.constrain_coefficients([0, 1], high=2.0, low=-1.0)(fit_method)(X, y)
    assert coefs[0] == 2.0 and coefs[1] == -1.0
    assert abs(coefs[5] - 1.0) < 0.5


def test_reference_r2_to_cv_error_on_our_predictions(ref_tools):
    from _oracle_backend import OracleBackend
    from sklearn.metrics import r2_score
    from sparselm_amd import _backend
    from sparselm_amd.model import Lasso

    rng = np.random.default_rng(1)
    X = rng.standard_normal((80, 5))
    y = X @ np.arange(1.0, 6.0) + rng.standard_normal(80)
    with _backend.use_backend(OracleBackend()):
        pred = Lasso(alpha=0.05, fit_intercept=True).fit(X, y).predict(X)
    err = ref_tools.r2_score_to_cv_error(r2_score(y, pred), y, pred)
    assert np.isfinite(err) and err >= 0
