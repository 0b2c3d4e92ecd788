"""Stepwise (piece-wise) fitting: a chain of estimators, each fitted on its own block of features
to the residual left by the previous ones.

Behavioural counterpart of the reference's ``StepwiseEstimator`` (src/sparselm/stepwise.py:43-237),
written for scikit-learn >= 1.6 (the reference's version relies on ``_validate_data``, removed from
scikit-learn): same constructor, same parameter-routing through ``steps``, same three structural
rules (feature blocks form a partition of ``range(n_features)``; only the first step may fit an
intercept; no nesting), same fitted attributes (``coef_`` assembled from the blocks, ``intercept_``
summed over steps).  Pure composition above the fit path: any estimator exposing ``fit`` / ``coef_`` /
``intercept_`` / ``fit_intercept`` works, including the searchers of ``sparselm_amd.model_selection``.
"""

from __future__ import annotations

import numpy as np
from sklearn.base import RegressorMixin
from sklearn.utils._param_validation import InvalidParameterError
from sklearn.utils.metaestimators import _BaseComposition
from sklearn.utils.validation import _check_sample_weight, check_is_fitted, validate_data

__all__ = ["StepwiseEstimator"]


def _inner(estimator):
    """The regressor a step finally delegates to (a searcher exposes it as ``estimator``)."""
    return estimator.estimator if hasattr(estimator, "estimator") else estimator


def _fitted(estimator):
    check_is_fitted(estimator)
    model = estimator.best_estimator_ if hasattr(estimator, "best_estimator_") else estimator
    if not hasattr(model, "coef_"):
        raise ValueError(f"Estimator {estimator} is not a valid linear model!")
    return model


class StepwiseEstimator(_BaseComposition, RegressorMixin):
    """Composite regressor fitted block by block on residuals.

    Args:
        steps (list[tuple[str, estimator]]): named estimators, applied in order.
        estimator_feature_indices (tuple[tuple[int]]): feature columns owned by each step.  Group
            labels / hierarchies of a step must already refer to its own sliced feature block.
    """

    def __init__(self, steps, estimator_feature_indices):
        self.steps = steps
        self.estimator_feature_indices = estimator_feature_indices

    def get_params(self, deep=True):
        return self._get_params("steps", deep=deep)

    def set_params(self, **params):
        self._set_params("steps", **params)
        return self

    def _check_structure(self):
        flat = sorted(i for scope in self.estimator_feature_indices for i in scope)
        if flat != list(range(len(flat))):
            raise InvalidParameterError(
                f"Given feature indices: {self.estimator_feature_indices} are not continuous and "
                "non-overlapping series starting from 0!"
            )
        if len(self.steps) != len(self.estimator_feature_indices):
            raise InvalidParameterError("steps and estimator_feature_indices must have the same length")
        for k, (_, est) in enumerate(self.steps):
            if isinstance(est, StepwiseEstimator):
                raise InvalidParameterError("StepwiseEstimator should not be nested with another StepwiseEstimator!")
            if k > 0 and getattr(_inner(est), "fit_intercept", False):
                raise InvalidParameterError("Only the first estimator in steps is allowed to fit intercept!")
        return len(flat)

    def fit(self, X, y, sample_weight=None, *args, **kwargs):
        """Fit every step on ``X[:, scope]`` against the running residual."""
        n_features = self._check_structure()
        X, y = validate_data(self, X, y, accept_sparse=False, ensure_2d=True, y_numeric=True, multi_output=True)
        if X.shape[1] != n_features:
            raise ValueError(f"X has {X.shape[1]} features, the steps cover {n_features}")
        if sample_weight is not None:
            sample_weight = _check_sample_weight(sample_weight, X, dtype=X.dtype)
        residual = np.array(y, dtype=np.float64)
        coef = np.full(X.shape[1], np.nan)
        intercept = 0.0
        for (_, est), scope in zip(self.steps, self.estimator_feature_indices):
            cols = list(scope)
            if sample_weight is None:
                est.fit(X[:, cols], residual, *args, **kwargs)
            else:
                est.fit(X[:, cols], residual, *args, sample_weight=sample_weight, **kwargs)
            model = _fitted(est)
            coef[cols] = model.coef_
            intercept = intercept + model.intercept_
            residual = residual - est.predict(X[:, cols])
        self.coef_ = coef
        self.intercept_ = intercept
        return self

    def predict(self, X):
        check_is_fitted(self, "coef_")
        X = validate_data(self, X, accept_sparse=False, reset=False)
        return X @ self.coef_ + self.intercept_
