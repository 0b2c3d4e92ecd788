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

__all__ = ["OrdinaryLeastSquares", "Lasso", "GroupLasso", "OverlapGroupLasso", "SparseGroupLasso", "RidgedGroupLasso"]


class OrdinaryLeastSquares(ProxRegressor):
    r"""Ordinary least squares ``1/(2n)||X b - y||^2`` (reference model/_ols.py:16-65): the penalty-free
    member of the family, solved by the same engine."""

    def _penalty(self, X):
        return None, None, None, None, X.shape[1]


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
        standardize (bool): penalise ``||X_g b_g||`` instead of ``||b_g||`` (reference :249-252): solved
            through a per-group change of variables (``standardize_groups``); rank-deficient groups get the
            minimum-norm coefficients among those the objective cannot tell apart.
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

    def _needs_host_preprocessing(self) -> bool:
        return bool(self.standardize)  # the per-group QR works on the centred design

    def _design_transform(self, X):
        if not self.standardize:
            return Design(X)
        gidx, G = dense_group_index(self.groups, X.shape[1])
        return standardize_groups(X, gidx, G)

    def _group_setup(self, X):
        gidx, G = dense_group_index(self.groups, X.shape[1])
        w = np.ones(G) if self.group_weights is None else np.asarray(self.group_weights, dtype=np.float64)
        return gidx, G, w

    def _penalty(self, X):
        gidx, G, w = self._group_setup(X)
        return None, self.alpha * w, None, gidx, G


class Design:
    """The design the engine is given for one fit, with the maps between its unknowns and the estimator's
    coefficients (identity unless ``standardize=True``).

    ``X``: the design; ``back(gamma) -> beta``; ``forward(beta) -> gamma`` (warm starts); ``extra_rows``
    rows of ``X`` carry a quadratic term instead of observations -- the targets there are zero -- and
    ``scale`` = sqrt(rows / n) multiplies design and targets so that the engine's 1/(2 rows) loss equals the
    reference's 1/(2 n); ``ridge_absorbed``: the ridge term lives in those rows, the penalty's ``d`` is dropped.
    """

    def __init__(self, X, back=None, forward=None, extra_rows=0, scale=1.0, ridge_absorbed=False):
        self.X, self._back, self._forward = X, back, forward
        self.extra_rows, self.scale, self.ridge_absorbed = int(extra_rows), float(scale), bool(ridge_absorbed)

    @property
    def identity(self):
        return self._back is None

    def back(self, gamma):
        return gamma if self._back is None else self._back(gamma)

    def warm(self, beta):
        if beta is None or self._forward is None:
            return beta if self._back is None else None
        return self._forward(beta)

    def target(self, y):
        if not self.extra_rows:
            return y
        return self.scale * np.concatenate([np.asarray(y, dtype=np.float64), np.zeros(self.extra_rows)])


def _thin_svd(A):
    """Thin SVD of a tall block (a group's columns: n x |g|, |g| << n).  Through the |g| x |g| Gram matrix -- eigenvectors
    V, singular values sqrt(lambda), U = A V / s -- where that keeps U orthonormal to ~1e-12: singular values within 1e2
    of each other (the Gram squares the condition number: at a ratio of 1e4 the columns of U were measured 2e-9 away from
    orthonormal and the singular values 1e-9 off, next to the solver's tolerance), LAPACK's SVD of the block itself
    otherwise (rank-deficient or ill-conditioned groups, blocks that are not tall).  A group of 20 000 x 10: 0.3 ms instead
    of 2.5 -- the 500 groups of a 100 000 x 5 000 design are 0.7 s of host time instead of 5."""
    n, k = A.shape
    if k == 0 or n < 4 * k:
        return np.linalg.svd(A, full_matrices=False)
    lam, V = np.linalg.eigh(A.T @ A)
    lam, V = lam[::-1].copy(), np.ascontiguousarray(V[:, ::-1])  # descending, as the SVD orders them
    if not (lam[-1] > 1e-4 * lam[0] > 0.0):
        return np.linalg.svd(A, full_matrices=False)
    sv = np.sqrt(lam)
    return (A @ V) / sv, sv, V.T


def standardize_groups(X, gidx, n_groups, delta=None):
    """``standardize=True`` as a per-group change of variables that turns the penalised quantity into an
    ordinary group norm, so the problem stays inside the prox family.

    * ``delta is None`` -- GroupLasso and its adaptive / overlap variants: the penalty is ``||X_g beta_g||_2``
      (reference _lasso.py:249-252).  Thin SVD ``X_g = U_g S_g V_g^T``; with ``gamma_g = S_g V_g^T beta_g`` the
      penalty is ``||gamma_g||_2`` on the design ``U`` and the loss is unchanged (``X beta = U gamma``).  A
      rank-deficient group (duplicated columns, more features than samples) keeps as many unknowns as its
      rank -- the other columns of the design are zero and stay at zero -- and ``back`` returns the
      minimum-norm ``beta_g = V_g S_g^-1 gamma_g`` among the coefficient vectors with that ``X_g beta_g``
      (the objective does not tell them apart).
    * ``delta`` given -- RidgedGroupLasso (reference _lasso.py:767-793): the penalty is
      ``||M_g beta_g||_2`` with ``M_g = sqrtm(X_g^T X_g + sqrt(delta_g) I)`` and the ridge ``1/2 delta_g
      ||beta_g||^2`` stays on ``beta``.  With ``gamma_g = M_g beta_g`` the design is ``X_g M_g^-1`` and the ridge
      becomes the quadratic ``1/2 delta_g ||M_g^-1 gamma_g||^2`` -- carried as ``|g|`` extra rows
      ``sqrt(n delta_g) M_g^-1`` with zero targets under the same 1/(2n) loss.

    Returns a ``Design``."""
    X = np.asarray(X, dtype=np.float64)
    n, p = X.shape
    gidx = np.arange(p) if gidx is None else np.asarray(gidx)
    factors = []
    if delta is None:
        Q = np.zeros_like(X)
        for g in range(n_groups):
            cols = np.flatnonzero(gidx == g)
            if not len(cols):
                continue
            u, sv, vt = _thin_svd(X[:, cols])
            r = int(np.sum(sv > 1e-12 * max(sv[0], 1e-300))) if len(sv) else 0
            Q[:, cols[:r]] = u[:, :r]
            factors.append((cols, r, sv[:r], vt[:r]))

        def back(gamma):
            beta = np.zeros_like(gamma)
            for cols, r, sv, vt in factors:
                beta[cols] = vt.T @ (gamma[cols[:r]] / sv)
            return beta

        def forward(beta):
            gamma = np.zeros_like(beta)
            for cols, r, sv, vt in factors:
                gamma[cols[:r]] = sv * (vt @ beta[cols])
            return gamma

        return Design(Q, back, forward)

    delta = np.broadcast_to(np.asarray(delta, dtype=np.float64), (n_groups,))
    rows = n + p
    Xa = np.zeros((rows, p))
    for g in range(n_groups):
        cols = np.flatnonzero(gidx == g)
        if not len(cols):
            continue
        lam, V = np.linalg.eigh(X[:, cols].T @ X[:, cols])
        m = np.sqrt(np.maximum(lam, 0.0) + np.sqrt(delta[g]))
        if m.min() <= 1e-12 * max(m.max(), 1e-300):
            raise ValueError(f"standardize=True: group {g} is rank deficient and delta is zero there: "
                             "sqrtm(X_g^T X_g + sqrt(delta_g) I) is singular")
        Minv = (V / m) @ V.T
        Xa[:n, cols] = X[:, cols] @ Minv
        Xa[n + cols[:, None], cols[None, :]] = np.sqrt(n * delta[g]) * Minv
        factors.append((cols, Minv, (V * m) @ V.T))
    scale = np.sqrt(rows / n)

    def back(gamma):
        beta = np.zeros_like(gamma)
        for cols, Minv, _ in factors:
            beta[cols] = Minv @ gamma[cols]
        return beta

    def forward(beta):
        gamma = np.zeros_like(beta)
        for cols, _, M in factors:
            gamma[cols] = M @ beta[cols]
        return gamma

    return Design(scale * Xa, back, forward, extra_rows=p, scale=scale, ridge_absorbed=True)


def overlap_extension(group_list, n_features):
    """Column duplication that makes overlapping groups disjoint (reference _lasso.py:440-461):
    returns (beta_indices, extended_groups); ``None`` means singleton groups (:446-448)."""
    if group_list is None:
        group_list = [[i] for i in range(n_features)]
    group_ids = np.sort(np.unique([gid for grp in group_list for gid in grp]))
    inds = [[i for i, grp in enumerate(group_list) if gid in grp] for gid in group_ids]
    extended_groups = np.concatenate([len(g) * [i] for i, g in enumerate(inds)])
    return np.concatenate(inds).astype(np.int64), extended_groups.astype(np.int32)


class OverlapGroupLasso(GroupLasso):
    r"""Overlap Group Lasso: ``alpha sum_G w_G ||b_G||_2`` where a coefficient may belong to several
    groups (reference _lasso.py:279-502).  Solved, as in the reference, as an ordinary group lasso on
    the design with duplicated columns ``X[:, beta_indices]`` (:440-461); the coefficients of the
    copies are summed back (:486-502).

    Args:
        group_list (list[list[int]] | None): for every feature the ids of the groups it belongs to;
            ``None`` warns and reduces to a Lasso (:398-404).
    """

    def __init__(
        self,
        group_list=None,
        alpha=1.0,
        group_weights=None,
        standardize=False,
        fit_intercept=False,
        copy_X=True,
        warm_start=False,
        solver=None,
        solver_options=None,
    ):
        self.group_list = group_list
        GroupLasso.__init__(
            self,
            groups=None,
            alpha=alpha,
            group_weights=group_weights,
            standardize=standardize,
            fit_intercept=fit_intercept,
            copy_X=copy_X,
            warm_start=warm_start,
            solver=solver,
            solver_options=solver_options,
        )

    @classmethod
    def _get_param_names(cls):
        # `groups` is fixed to None by the constructor and is not a hyper-parameter of this class
        return sorted(n for n in super()._get_param_names() if n != "groups")

    def _needs_host_preprocessing(self) -> bool:
        return True  # the duplicated-column design is assembled on the host

    def _n_groups(self, n_features):
        if self.group_list is None:
            return n_features
        return len(np.unique([gid for grp in self.group_list for gid in grp]))

    def _validate_params(self, X, y) -> None:
        """Reference _lasso.py:384-406 (skips GroupLasso's own group checks)."""
        Lasso._validate_params(self, X, y)
        if self.group_list is not None:
            if len(self.group_list) != X.shape[1]:
                raise ValueError("The length of the group list must be the same as the number of features.")
        else:
            warnings.warn(
                "No group list has been supplied such that the problem reduces to"
                " a simple Lasso. You should consider using that instead.",
                UserWarning,
            )
        check_group_weights(self.group_weights, self._n_groups(X.shape[1]))

    def _extended(self, X):
        bidx, ext = overlap_extension(self.group_list, X.shape[1])
        return bidx, ext, int(ext.max()) + 1

    def _solve(self, X, y, solver_options, *args, **kwargs):
        from .._backend import get_backend

        p = X.shape[1]
        bidx, ext, G = self._extended(X)
        w = np.ones(G) if self.group_weights is None else np.asarray(self.group_weights, dtype=np.float64)
        X_ext = np.ascontiguousarray(X[:, bidx])
        dz = standardize_groups(X_ext, ext, G) if self.standardize else Design(X_ext)
        problem = self._open_problem(dz.X, dz.target(y), ext, G, solver_options)
        try:
            beta_ext, _, info = problem.solve(np.zeros(len(bidx)), self.alpha * w, np.zeros(G))
        finally:
            problem.close()
        self.solver_info_ = info
        return np.bincount(bidx, weights=dz.back(beta_ext), minlength=p)


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

    # standardize=True: lambda2 sum_g w_g ||X_g b_g||_2 next to an l1 norm of b itself (reference :627-639 with
    # :249-252) -- no change of variables makes both separable, so the design stays as it is and the solve is a
    # splitting around weighted-l1 engine solves (model/_split.py)
    def _design_transform(self, X):
        return Design(X)

    def _open_problem(self, X, y, gidx, G, solver_options):
        if not self.standardize:
            return super()._open_problem(X, y, gidx, G, solver_options)
        from ._split import StandardizedSparseGroupProblem

        return StandardizedSparseGroupProblem(X, y, gidx, G, solver_options)

    def _penalty(self, X):
        gidx, G, w = self._group_setup(X)
        lam1, lam2 = self._lambdas()
        return lam1 * np.ones(X.shape[1]), lam2 * w, None, gidx, G


class RidgedGroupLasso(GroupLasso):
    r"""Ridged Group Lasso: ``alpha sum_g w_g ||b_g||_2 + 1/2 sum_g delta_g ||b_g||_2^2``
    (reference _lasso.py:642-811; ridge term :795-811, not scaled by n).

    Args:
        delta (ndarray | tuple): ridge weight, length 1 (shared) or one per group (:744-765).
        standardize (bool): the group norms become ``||sqrtm(X_g^T X_g + sqrt(delta_g) I) b_g||_2``
            (reference :767-793) while the ridge stays on ``b``; see ``standardize_groups(delta=...)``.
    """

    _hyper_parameter_constraints: dict = {
        "alpha": [Interval(type=Real, left=0.0, right=None, closed="left")],
        "delta": ["array-like", Interval(type=Real, left=0.0, right=None, closed="left")],
    }

    def _design_transform(self, X):
        if not self.standardize:
            return Design(X)
        gidx, G = dense_group_index(self.groups, X.shape[1])
        return standardize_groups(X, gidx, G, delta=self._delta_vector(G))

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
