"""Base regressor of the HIP-backed estimators.

Mirrors ``CVXRegressor`` (reference src/sparselm/model/_base.py:69-519) for everything a user sees:
constructor arguments (:128-140), ``fit`` flow (:142-205), preprocessing (:207-227), parameter
validation through sklearn's declarative constraints (:229-245), ``intercept_`` convention.  The
cvxpy problem objects (``canonicals_``, ``generate_problem``, ``add_constraints``) have no meaning
for a proximal-gradient engine and are not provided.
"""

from __future__ import annotations

import warnings
import numpy as np
from sklearn.base import BaseEstimator, RegressorMixin
from sklearn.utils._param_validation import validate_parameter_constraints
from sklearn.utils.validation import _check_sample_weight, check_is_fitted, validate_data

from .._backend import get_backend, normalise_options

_DEVICE_SCAN_FROM = 1 << 20  # entries of X from which `fit` leaves the NaN / infinity scan to the device copy

__all__ = ["ProxRegressor"]


class ProxRegressor(RegressorMixin, BaseEstimator):
    """Linear model fitted by minimising ``1/(2n)||Xb - y||^2 + penalty(b)`` on the GPU.

    Sub-classes define the penalty through ``_penalty(X)``: vectors ``(a, b, d)`` of the family
    ``sum_j a_j|b_j| + sum_g b_g||b_g||_2 + 1/2 sum_g d_g||b_g||_2^2`` and the group index.

    Args:
        fit_intercept (bool): centre X and y by their (weighted) means and report ``intercept_``.
        copy_X (bool): kept for signature compatibility; X is never modified in place.
        warm_start (bool): start from the previous ``coef_`` when re-fitting.
        solver (str | None): ``None`` or ``"hip"``.  cvxpy solver names are accepted and ignored
            with a warning so reference scripts run unchanged.
        solver_options (dict | None): engine options: ``tol`` -- the relative distance to the minimiser a
            solution is accepted at (KKT residual <= tol * mu * ||coef||, mu the strong-convexity estimate of
            the active face; default 1e-10 below 2^26 matrix entries, 1e-8 above: ``_backend.default_tol``),
            ``max_iter`` (10000), ``L`` (Lipschitz constant, default estimated), ``restart`` (True),
            ``check_every``, ``device``, ``on_chip`` (False: never the one-workgroup solvers), ``covariance`` (True: passes
            from the Gram of the rows instead of X -- built once per device dataset; "auto", the default, lets a
            ``GridSearchCV`` decide and a single fit decline).

    Attributes:
        coef_ (ndarray of shape (n_features,)), intercept_ (float), solver_info_ (dict).
    """

    _parameter_constraints: dict = {
        "fit_intercept": ["boolean"],
        "copy_X": ["boolean"],
        "warm_start": ["boolean"],
        "solver": [str, None],
        "solver_options": [dict, None],
    }
    # constraints on the regularisation hyper-parameters (the reference's
    # ``_cvx_parameter_constraints``, _base.py:126)
    _hyper_parameter_constraints: dict | None = None

    def __init__(self, fit_intercept=False, copy_X=True, warm_start=False, solver=None, solver_options=None):
        self.fit_intercept = fit_intercept
        self.copy_X = copy_X
        self.warm_start = warm_start
        self.solver = solver
        self.solver_options = solver_options

    @classmethod
    def _get_param_names(cls):
        # sklearn reads the constructor's signature with `inspect` on EVERY get_params / clone / set_params (35 us a
        # time; a grid search over 50 cells asks 200 times, a reference-sized fit spends a third of its time there):
        # the names of a class do not change, so they are kept with the class (its own __dict__: not inherited).
        names = cls.__dict__.get("_param_names_of_class")
        if names is None:
            names = list(super()._get_param_names())
            cls._param_names_of_class = names
        return list(names)

    # ---------------------------------------------------------------------------------------
    def fit(self, X, y, sample_weight=None, *args, **kwargs):
        """Fit the coefficients (reference flow: _base.py:142-205)."""
        # (validation as in the reference, with one difference of order: on the device route the scan of a LARGE X for NaN /
        #  infinity -- numpy's sum over the array in check_array: 0.145 s for a 100 000 x 5 000 design, two thirds of a whole
        #  fit -- is made on the device copy right after the upload (_backend.raise_if_nonfinite: the same ValueError).  y,
        #  shapes, dtypes and small arrays are checked here as always.)
        native = getattr(get_backend(), "native_preprocessing", False) and not self._needs_host_preprocessing()
        defer_scan = bool(native and getattr(X, "size", 0) >= _DEVICE_SCAN_FROM)
        X, y = validate_data(self, X, y, accept_sparse=False, y_numeric=True, multi_output=False,
                             ensure_all_finite=not defer_scan)
        # Preprocessing (reference _base.py:207-227).  With the HIP backend the same arithmetic runs
        # on the device: normalised sample weights become row weights of the fused kernel, centring is
        # done in place on the engine's copy -- the host never builds a second X.
        self._native = None
        if native:
            X = np.asarray(X, dtype=np.float64)
            y = np.asarray(y, dtype=np.float64)
            w = None
            if sample_weight is not None:
                w = _check_sample_weight(sample_weight, X, dtype=X.dtype)
                w = w * (X.shape[0] / np.sum(w))
            self._native = {"row_weight": w, "center": bool(self.fit_intercept), "check_finite": defer_scan}
            X_offset, y_offset = np.zeros(X.shape[1]), 0.0
        else:
            X, y, X_offset, y_offset = self._preprocess_data(X, y, sample_weight)
        self._validate_params(X, y)

        solver_options = self.solver_options if self.solver_options is not None else {}
        if not isinstance(solver_options, dict):
            raise TypeError("solver_options must be a dictionary")
        if isinstance(self.solver, str) and self.solver.lower() not in ("hip", "fista"):
            warnings.warn(
                f"solver={self.solver!r} names a cvxpy back-end; sparselm_amd always uses its HIP "
                "FISTA engine and ignores it.",
                UserWarning,
            )
        try:
            self.coef_ = self._solve(X, y, normalise_options(solver_options), *args, **kwargs)
            if self._native is not None and self._native.get("offsets") is not None:
                X_offset, y_offset = self._native["offsets"]
        finally:
            native, self._native = self._native, None
        del native
        self._set_intercept(X_offset, y_offset)
        return self

    def predict(self, X):
        check_is_fitted(self)
        X = validate_data(self, X, accept_sparse=False, reset=False)
        return X @ self.coef_ + self.intercept_

    # ---------------------------------------------------------------------------------------
    def _preprocess_data(self, X, y, sample_weight=None):
        """Same recipe as reference _base.py:207-227 (+ sklearn's _preprocess_data/_rescale_data):
        weights rescaled to sum to n, (weighted) centring iff fit_intercept, rows times sqrt(w)."""
        X = np.asarray(X, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        n = X.shape[0]
        if sample_weight is not None:
            sample_weight = _check_sample_weight(sample_weight, X, dtype=X.dtype)
            sample_weight = sample_weight * (n / np.sum(sample_weight))
        if self.fit_intercept:
            X_offset = np.average(X, axis=0, weights=sample_weight)
            y_offset = np.average(y, axis=0, weights=sample_weight)
            X = X - X_offset
            y = y - y_offset
        else:
            X_offset = np.zeros(X.shape[1], dtype=X.dtype)
            y_offset = 0.0
        if sample_weight is not None:
            sw = np.sqrt(sample_weight)
            X = X * sw[:, None]
            y = y * sw
        return X, y, X_offset, y_offset

    def _set_intercept(self, X_offset, y_offset):
        if self.fit_intercept:
            self.intercept_ = float(y_offset - np.dot(X_offset, self.coef_))
        else:
            self.intercept_ = 0.0

    def _validate_params(self, X, y) -> None:
        """Declarative hyper-parameter validation (reference _base.py:229-245)."""
        constraints = dict(self._parameter_constraints)
        if self._hyper_parameter_constraints is not None:
            constraints.update(self._hyper_parameter_constraints)
        validate_parameter_constraints(
            constraints, self.get_params(deep=False), caller_name=self.__class__.__name__
        )

    # ---------------------------------------------------------------------------------------
    def _penalty(self, X):
        """Return (a, b, d, gidx, n_groups); a: (p,) or None (=0), b/d: (G,) or None (=0)."""
        raise NotImplementedError

    def _needs_host_preprocessing(self) -> bool:
        """True when ``_solve`` manipulates the preprocessed design on the host (column duplication,
        per-group QR) and therefore needs the centred / re-weighted X itself."""
        return False

    def _open_problem(self, X, y, gidx, G, solver_options):
        """Upload (X, y) to the backend; in native mode with device-side weights and centring."""
        native = getattr(self, "_native", None)
        if native is None:
            return get_backend().problem(X, y, gidx, G, solver_options)
        problem = get_backend().problem(
            X, y, gidx, G, solver_options, row_weight=native["row_weight"], center=native["center"],
            **({"check_finite": True} if native.get("check_finite") else {})
        )
        if native["center"]:
            native["offsets"] = (problem.x_mean, problem.y_mean)
        return problem

    def _design_transform(self, X):
        """Hook: the design the engine is given and the maps between its unknowns and the coefficients of X
        (a ``_lasso.Design``; the identity here).  Used by ``standardize=True``."""
        from ._lasso import Design

        return Design(X)

    def _warm_beta(self, n_features):
        if self.warm_start and hasattr(self, "coef_") and np.shape(self.coef_) == (n_features,):
            if np.all(np.isfinite(self.coef_)):
                return np.asarray(self.coef_, dtype=np.float64)
        return None

    def _solve(self, X, y, solver_options, *args, **kwargs):
        """Counterpart of CVXRegressor._solve (reference _base.py:512-519): one minimisation."""
        a, b, d, gidx, G = self._penalty(X)
        p = X.shape[1]
        dz = self._design_transform(X)
        problem = self._open_problem(dz.X, dz.target(y), gidx, G, solver_options)
        try:
            beta, _, info = problem.solve(
                np.zeros(p) if a is None else a,
                np.zeros(G) if b is None else b,
                np.zeros(G) if (d is None or dz.ridge_absorbed) else d,
                beta0=dz.warm(self._warm_beta(p)),
            )
        finally:
            problem.close()
        self.solver_info_ = info
        return dz.back(beta)

    def __sklearn_tags__(self):
        tags = super().__sklearn_tags__()
        tags.target_tags.single_output = True
        return tags

