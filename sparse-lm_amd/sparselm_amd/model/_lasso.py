"""Lasso, GroupLasso, SparseGroupLasso, RidgedGroupLasso on the HIP engine.

Same constructor signatures, validation, warnings and objective scaling as the reference
(src/sparselm/model/_lasso.py).  Objective everywhere: ``1/(2 n_samples) ||X b - y||^2 + penalty``
(reference :109-121; the docstring formulas there omit the 1/(2n), the code is authoritative and
is pinned by tests/test_lasso.py:29-61).
"""

from __future__ import annotations

import warnings
from collections.abc import Sequence
from numbers import Real

import numpy as np
from sklearn.utils._param_validation import Interval
from sklearn.utils.validation import check_scalar

from .._utils.validation import check_group_weights, check_groups, dense_group_index
from ._base import ProxRegressor

__all__ = ["Lasso", "GroupLasso", "SparseGroupLasso", "RidgedGroupLasso"]


class Lasso(ProxRegressor):
    r"""Lasso: ``1/(2n)||X b - y||^2 + alpha ||b||_1``  (reference _lasso.py:34-121).

    Args:
        alpha (float): regularisation strength, >= 0.
        fit_intercept, copy_X, warm_start, solver, solver_options: see ``ProxRegressor``.
    """

    _hyper_parameter_constraints: dict = {"alpha": [Interval(type=Real, left=0.0, right=None, closed="left")]}

    def __init__(
        self, alpha=1.0, fit_intercept=False, copy_X=True, warm_start=False, solver=None, solver_options=None
    ):
        ProxRegressor.__init__(
            self,
            fit_intercept=fit_intercept,
            copy_X=copy_X,
            warm_start=warm_start,
            solver=solver,
            solver_options=solver_options,
        )
        self.alpha = alpha

    def _penalty(self, X):
        p = X.shape[1]
        return self.alpha * np.ones(p), None, None, None, p


class GroupLasso(Lasso):
    r"""Group Lasso: ``1/(2n)||X b - y||^2 + alpha sum_g w_g ||b_g||_2``  (reference _lasso.py:124-275).

    Args:
        groups (list | ndarray | None): group label of every feature, shape (n_features,).  The
            i-th sorted unique label is group i (reference :248).  ``None`` warns and treats each
            feature as its own group (:211-217).
        alpha (float): regularisation strength.
        group_weights (ndarray | None): weight per group; default ones (reference :233-235 -- the
            reference docstring says sqrt(group size) but the code uses ones).
        standardize (bool): penalise ``||X_g b_g||`` instead of ``||b_g||`` (reference :249-252).
            Not implemented by the HIP engine yet.
    """

    def __init__(
        self,
        groups=None,
        alpha=1.0,
        group_weights=None,
        standardize=False,
        fit_intercept=False,
        copy_X=True,
        warm_start=False,
        solver=None,
        solver_options=None,
    ):
        self.groups = groups
        self.standardize = standardize
        self.group_weights = group_weights
        Lasso.__init__(
            self,
            alpha=alpha,
            fit_intercept=fit_intercept,
            copy_X=copy_X,
            warm_start=warm_start,
            solver=solver,
            solver_options=solver_options,
        )

    def _validate_params(self, X, y) -> None:
        """Group checks in the reference's order (_lasso.py:208-222)."""
        super()._validate_params(X, y)
        if self.groups is None:
            warnings.warn(
                "groups has not been supplied such that the problem reduces to"
                " a simple Lasso. You should consider using that instead.",
                UserWarning,
            )
            n_groups = X.shape[1]
        else:
            n_groups = len(np.unique(self.groups))
        check_groups(self.groups, X.shape[1])
        check_group_weights(self.group_weights, n_groups)
        if self.standardize:
            raise NotImplementedError(
                "standardize=True (penalty on ||X_g b_g||) is not implemented by the HIP engine yet"
            )

    def _group_setup(self, X):
        gidx, G = dense_group_index(self.groups, X.shape[1])
        w = np.ones(G) if self.group_weights is None else np.asarray(self.group_weights, dtype=np.float64)
        return gidx, G, w

    def _penalty(self, X):
        gidx, G, w = self._group_setup(X)
        return None, self.alpha * w, None, gidx, G


class SparseGroupLasso(GroupLasso):
    r"""Sparse Group Lasso: ``lambda1 ||b||_1 + lambda2 sum_g w_g ||b_g||_2`` with
    ``lambda1 = l1_ratio * alpha`` and ``lambda2 = (1 - l1_ratio) * alpha`` (reference _lasso.py:505-639)."""

    def __init__(
        self,
        groups=None,
        l1_ratio=0.5,
        alpha=1.0,
        group_weights=None,
        standardize=False,
        fit_intercept=False,
        copy_X=True,
        warm_start=False,
        solver=None,
        solver_options=None,
    ):
        GroupLasso.__init__(
            self,
            groups=groups,
            alpha=alpha,
            group_weights=group_weights,
            standardize=standardize,
            fit_intercept=fit_intercept,
            copy_X=copy_X,
            warm_start=warm_start,
            solver=solver,
            solver_options=solver_options,
        )
        self.l1_ratio = l1_ratio

    def _validate_params(self, X, y):
        """l1_ratio must be a float in [0, 1]; 0.0 and 1.0 only warn (reference :594-608)."""
        super()._validate_params(X, y)
        check_scalar(self.l1_ratio, "l1_ratio", float, min_val=0, max_val=1)
        if self.l1_ratio == 0.0:
            warnings.warn(
                "It is more efficient to use GroupLasso directly than SparseGroupLasso with l1_ratio=0",
                UserWarning,
            )
        if self.l1_ratio == 1.0:
            warnings.warn(
                "It is more efficient to use Lasso directly than SparseGroupLasso with l1_ratio=1",
                UserWarning,
            )

    def _lambdas(self):
        return self.l1_ratio * self.alpha, (1.0 - self.l1_ratio) * self.alpha

    def _penalty(self, X):
        gidx, G, w = self._group_setup(X)
        lam1, lam2 = self._lambdas()
        return lam1 * np.ones(X.shape[1]), lam2 * w, None, gidx, G


class RidgedGroupLasso(GroupLasso):
    r"""Ridged Group Lasso: ``alpha sum_g w_g ||b_g||_2 + 1/2 sum_g delta_g ||b_g||_2^2``
    (reference _lasso.py:642-811; ridge term :795-811, not scaled by n).

    Args:
        delta (ndarray | tuple): ridge weight, length 1 (shared) or one per group (:744-765).
    """

    _hyper_parameter_constraints: dict = {
        "alpha": [Interval(type=Real, left=0.0, right=None, closed="left")],
        "delta": ["array-like", Interval(type=Real, left=0.0, right=None, closed="left")],
    }

    def __init__(
        self,
        groups=None,
        alpha=1.0,
        delta=(1.0,),
        group_weights=None,
        standardize=False,
        fit_intercept=False,
        copy_X=True,
        warm_start=False,
        solver=None,
        solver_options=None,
    ):
        GroupLasso.__init__(
            self,
            groups=groups,
            alpha=alpha,
            group_weights=group_weights,
            standardize=standardize,
            fit_intercept=fit_intercept,
            copy_X=copy_X,
            warm_start=warm_start,
            solver=solver,
            solver_options=solver_options,
        )
        self.delta = delta

    def _validate_params(self, X, y) -> None:
        super()._validate_params(X, y)
        n_groups = X.shape[1] if self.groups is None else len(np.unique(self.groups))
        if len(self.delta) != n_groups and len(self.delta) != 1:
            raise ValueError(
                f"delta must be an array of length 1 or equal to the number of groups {n_groups}."
            )
        if np.any(np.asarray(self.delta, dtype=np.float64) < 0):
            raise ValueError("delta must be non-negative")

    def _delta_vector(self, G):
        delta = self.delta
        if isinstance(delta, (np.ndarray, Sequence)) and len(delta) == 1:
            return float(np.asarray(delta, dtype=np.float64)[0]) * np.ones(G)
        return np.asarray(delta, dtype=np.float64)

    def _penalty(self, X):
        gidx, G, w = self._group_setup(X)
        return None, self.alpha * w, self._delta_vector(G), gidx, G
