"""Adaptive (iteratively re-weighted) Lasso-family estimators on the HIP engine.

Host-side re-weighting loops around the device solve, with exactly the reference's semantics
(src/sparselm/model/_adaptive_lasso.py):

* outer loop of ``max_iter`` solves; weights are updated after EVERY solve, the last included;
  early stop when ``||w_new - w_prev||_2 <= tol``; ``n_iter_`` counts solves; the returned
  coefficients are those of the last solve (:206-232);
* default update ``update(x, eps) = alpha / (|x| + eps)`` (:177-182) and weights
  ``alpha * update(...)`` (:196-204) -- alpha enters twice;
* AdaptiveGroupLasso's first solve uses ``alpha * ones(G)`` and ignores ``group_weights``
  (:343-362); later ``(alpha * w_g) * update(||b_g||, eps)`` (:364-374);
* AdaptiveSparseGroupLasso starts from ``lambda1 * 1`` / ``lambda2 * 1`` (:654-668) and updates
  ``lambda1 * update(b)``, ``(lambda2 * w_g) * update(||b_g||)`` (:712-726), convergence on the
  concatenation ``[group weights, coefficient weights]`` (:698-710).

X stays resident in HBM across the outer iterations; only the p (or G) weights travel each round,
and each inner solve is warm-started from the previous one when ``warm_start=True`` (the
reference's default for these classes, :121).
"""

from __future__ import annotations

import warnings
from numbers import Integral, Real

import numpy as np
from sklearn.utils._param_validation import Interval

from .._backend import get_backend
from ._lasso import GroupLasso, Lasso, OverlapGroupLasso, RidgedGroupLasso, SparseGroupLasso

__all__ = [
    "AdaptiveLasso",
    "AdaptiveGroupLasso",
    "AdaptiveOverlapGroupLasso",
    "AdaptiveSparseGroupLasso",
    "AdaptiveRidgedGroupLasso",
]


class AdaptiveLasso(Lasso):
    r"""Adaptive Lasso: ``1/(2n)||X b - y||^2 + ||w * b||_1`` with iteratively updated ``w``
    (reference _adaptive_lasso.py:45-232).

    Args:
        alpha (float): regularisation strength.
        max_iter (int): number of re-weighted solves (default 3).
        eps (float): stabiliser in the weight update (default 1e-6).
        tol (float): stop when the weights move less than this in l2 norm (default 1e-10).
        update_function (callable | None): ``f(values, eps) -> array``; default ``alpha/(|x|+eps)``.

    Attributes:
        n_iter_ (int): number of solves performed.
        adaptive_weights_ (ndarray): weights after the final update (the reference exposes them as
            ``canonicals_.parameters.adaptive_weights.value``).
    """

    _parameter_constraints: dict = {
        "tol": [Interval(type=Real, left=0.0, right=1.0, closed="both")],
        "max_iter": [Interval(type=Integral, left=0, right=None, closed="left")],
        "eps": [Interval(type=Real, left=0.0, right=1.0, closed="both")],
        "update_function": [callable, None],
        **Lasso._parameter_constraints,
    }

    def __init__(
        self,
        alpha=1.0,
        max_iter=3,
        eps=1e-6,
        tol=1e-10,
        update_function=None,
        fit_intercept=False,
        copy_X=True,
        warm_start=True,
        solver=None,
        solver_options=None,
    ):
        Lasso.__init__(
            self,
            alpha=alpha,
            fit_intercept=fit_intercept,
            copy_X=copy_X,
            warm_start=warm_start,
            solver=solver,
            solver_options=solver_options,
        )
        self._init_adaptive(max_iter, eps, tol, update_function)

    def _init_adaptive(self, max_iter, eps, tol, update_function):
        self.tol = tol
        self.max_iter = max_iter
        self.eps = eps
        self.update_function = update_function

    def _validate_params(self, X, y) -> None:
        super()._validate_params(X, y)
        if self.max_iter == 1:
            warnings.warn(
                "max_iter is set to 1. It should ideally be set > 1, otherwise consider "
                "using a non-adaptive Regressor",
                UserWarning,
            )

    def _get_update_function(self):
        if self.update_function is None:
            return lambda beta, eps: self.alpha / (abs(beta) + eps)
        return self.update_function

    # ---- hooks specialised by the group variants --------------------------------------------
    def _adaptive_setup(self, X):
        """Return (gidx, G, initial flat weight vector)."""
        p = X.shape[1]
        return None, p, self.alpha * np.ones(p)

    def _weights_to_penalty(self, weights, p, G):
        """flat weights -> (a, b, d)."""
        return weights, np.zeros(G), np.zeros(G)

    def _updated_weights(self, beta, group_norms):
        update = self._get_update_function()
        return self.alpha * np.asarray(update(beta, self.eps), dtype=np.float64)

    _needs_group_norms = False

    def _reweight_rule(self, p, G):
        """The default weight update in the engine's terms -- ``(coef_scale, group_scale, numerator, eps, tol, n_coef,
        n_group)`` of ``slm_solve_lanes_reweighted``: new ``a_j = coef_scale * (numerator / (|b_j| + eps))``, new ``b_g =
        group_scale[g] * (numerator / (||b_g|| + eps))``, the operations of ``_updated_weights`` in its order -- or
        ``None`` with a user's ``update_function``, which only the host loop can call.  After ``_adaptive_setup``."""
        if self.update_function is not None:
            return None
        return (float(self.alpha), None, float(self.alpha), float(self.eps), float(self.tol), int(p), 0)

    # ---- the re-weighting loop ---------------------------------------------------------------
    def _solve(self, X, y, solver_options, *args, **kwargs):
        """Counterpart of AdaptiveLasso._solve (reference _adaptive_lasso.py:206-232)."""
        p = X.shape[1]
        gidx, G, weights = self._adaptive_setup(X)
        previous_weights = weights.copy()
        # standardize=True: the loop runs in the per-group QR coordinates, where the engine's group norms
        # are ||X_g b_g|| -- what the reference feeds to the weight update (_adaptive_lasso.py:364-374)
        dz = self._design_transform(X)
        problem = self._open_problem(dz.X, dz.target(y), gidx, G, solver_options)
        beta = None
        warm = dz.warm(self._warm_beta(p)) if hasattr(self, "coef_") else None
        infos = []
        self.n_iter_ = 0
        try:
            # reference-sized problems with the default update function: every round inside one launch
            solve_rounds = getattr(problem, "solve_rounds", None)
            rule = self._reweight_rule(p, G) if solve_rounds is not None and self.max_iter >= 1 else None
            if rule is not None:
                a, b, d = self._weights_to_penalty(weights, p, G)
                got = solve_rounds(a, b, np.zeros(G) if dz.ridge_absorbed else d, rule, self.max_iter, beta0=warm, cold=not self.warm_start)
                if got is not None:
                    beta, group_norms, infos = got
                    self.n_iter_ = len(infos)
                    weights = self._updated_weights(beta, group_norms)
            for i in range(self.max_iter if beta is None else 0):
                a, b, d = self._weights_to_penalty(weights, p, G)
                if dz.ridge_absorbed:
                    d = np.zeros(G)
                beta0 = warm if beta is None else (beta if self.warm_start else None)
                beta, group_norms, info = problem.solve(
                    a, b, d, beta0=beta0, want_group_norms=self._needs_group_norms
                )
                infos.append(info)
                self.n_iter_ = i + 1
                weights = self._updated_weights(beta, group_norms)
                if np.linalg.norm(weights - previous_weights) <= self.tol:
                    break
                previous_weights = weights.copy()
        finally:
            problem.close()
        if beta is None:
            # the reference returns beta.value == None here (max_iter=0): make that an explicit error
            raise ValueError("max_iter=0 performs no solve; coef_ would be undefined")
        self.adaptive_weights_ = weights
        self.solver_info_ = {"solves": infos}
        return dz.back(beta)


class AdaptiveGroupLasso(AdaptiveLasso, GroupLasso):
    r"""Adaptive Group Lasso: ``sum_g w_g ||b_g||_2`` with iteratively updated group weights
    (reference _adaptive_lasso.py:235-374)."""

    def __init__(
        self,
        groups=None,
        alpha=1.0,
        group_weights=None,
        max_iter=3,
        eps=1e-6,
        tol=1e-10,
        update_function=None,
        standardize=False,
        fit_intercept=False,
        copy_X=True,
        warm_start=True,
        solver=None,
        solver_options=None,
    ):
        # explicit base initialisers instead of the reference's cooperative **kwargs chain
        # (sklearn >= 1.6 estimator checks reject a **kwargs constructor)
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
        self._init_adaptive(max_iter, eps, tol, update_function)

    _needs_group_norms = True

    def _adaptive_setup(self, X):
        gidx, G, w = self._group_setup(X)
        self._gw = w
        return gidx, G, self.alpha * np.ones(G)  # group_weights NOT applied in the first solve (:347-351)

    def _weights_to_penalty(self, weights, p, G):
        return np.zeros(p), weights, np.zeros(G)

    def _updated_weights(self, beta, group_norms):
        update = self._get_update_function()
        return (self.alpha * self._gw) * np.asarray(update(group_norms, self.eps), dtype=np.float64)

    def _reweight_rule(self, p, G):
        if self.update_function is not None:
            return None
        return (0.0, self.alpha * self._gw, float(self.alpha), float(self.eps), float(self.tol), 0, int(G))


class AdaptiveOverlapGroupLasso(AdaptiveGroupLasso, OverlapGroupLasso):
    r"""Adaptive Overlap Group Lasso (reference _adaptive_lasso.py:377-524): the AdaptiveGroupLasso
    re-weighting loop on the column-duplicated design, coefficients folded back (:515-524)."""

    def __init__(
        self,
        group_list=None,
        alpha=1.0,
        group_weights=None,
        max_iter=3,
        eps=1e-6,
        tol=1e-10,
        update_function=None,
        standardize=False,
        fit_intercept=False,
        copy_X=True,
        warm_start=True,
        solver=None,
        solver_options=None,
    ):
        OverlapGroupLasso.__init__(
            self,
            group_list=group_list,
            alpha=alpha,
            group_weights=group_weights,
            standardize=standardize,
            fit_intercept=fit_intercept,
            copy_X=copy_X,
            warm_start=warm_start,
            solver=solver,
            solver_options=solver_options,
        )
        self._init_adaptive(max_iter, eps, tol, update_function)

    def _validate_params(self, X, y) -> None:
        OverlapGroupLasso._validate_params(self, X, y)
        if self.max_iter == 1:
            warnings.warn(
                "max_iter is set to 1. It should ideally be set > 1, otherwise consider "
                "using a non-adaptive Regressor",
                UserWarning,
            )

    def _solve(self, X, y, solver_options, *args, **kwargs):
        p = X.shape[1]
        bidx, ext, G = self._extended(X)
        self._ext_groups = ext
        beta_ext = AdaptiveLasso._solve(self, np.ascontiguousarray(X[:, bidx]), y, solver_options)
        return np.bincount(bidx, weights=beta_ext, minlength=p)

    def _design_transform(self, X_ext):
        from ._lasso import Design, standardize_groups

        if not self.standardize:
            return Design(X_ext)
        return standardize_groups(X_ext, self._ext_groups, int(self._ext_groups.max()) + 1)

    def _adaptive_setup(self, X_ext):
        G = int(self._ext_groups.max()) + 1
        self._gw = np.ones(G) if self.group_weights is None else np.asarray(self.group_weights, dtype=np.float64)
        return self._ext_groups, G, self.alpha * np.ones(G)

    def _warm_beta(self, n_features):
        return None  # coef_ lives in the folded space; the extended problem starts cold


class AdaptiveSparseGroupLasso(AdaptiveLasso, SparseGroupLasso):
    r"""Adaptive Sparse Group Lasso (reference _adaptive_lasso.py:527-726)."""

    def __init__(
        self,
        groups=None,
        l1_ratio=0.5,
        alpha=1.0,
        group_weights=None,
        max_iter=3,
        eps=1e-6,
        tol=1e-10,
        update_function=None,
        standardize=False,
        fit_intercept=False,
        copy_X=True,
        warm_start=True,
        solver=None,
        solver_options=None,
    ):
        SparseGroupLasso.__init__(
            self,
            groups=groups,
            l1_ratio=l1_ratio,
            alpha=alpha,
            group_weights=group_weights,
            standardize=standardize,
            fit_intercept=fit_intercept,
            copy_X=copy_X,
            warm_start=warm_start,
            solver=solver,
            solver_options=solver_options,
        )
        self._init_adaptive(max_iter, eps, tol, update_function)

    _needs_group_norms = True

    def _adaptive_setup(self, X):
        gidx, G, w = self._group_setup(X)
        self._gw = w
        lam1, lam2 = self._lambdas()
        # concatenation order [group weights, coefficient weights] as in :686-696
        return gidx, G, np.concatenate((lam2 * np.ones(G), lam1 * np.ones(X.shape[1])))

    def _weights_to_penalty(self, weights, p, G):
        return weights[G:], weights[:G], np.zeros(G)

    def _updated_weights(self, beta, group_norms):
        update = self._get_update_function()
        lam1, lam2 = self._lambdas()
        coef_w = lam1 * np.asarray(update(beta, self.eps), dtype=np.float64)
        group_w = (lam2 * self._gw) * np.asarray(update(group_norms, self.eps), dtype=np.float64)
        return np.concatenate((group_w, coef_w))

    def _reweight_rule(self, p, G):
        if self.update_function is not None:
            return None
        lam1, lam2 = self._lambdas()
        return (float(lam1), lam2 * self._gw, float(self.alpha), float(self.eps), float(self.tol), int(p), int(G))

    @property
    def adaptive_group_weights_(self):
        G = len(self._gw)
        return self.adaptive_weights_[:G]

    @property
    def adaptive_coef_weights_(self):
        G = len(self._gw)
        return self.adaptive_weights_[G:]


class AdaptiveRidgedGroupLasso(AdaptiveGroupLasso, RidgedGroupLasso):
    r"""Adaptive Ridged Group Lasso (reference _adaptive_lasso.py:729-860): adaptive group weights
    plus the fixed ridge term ``1/2 sum_g delta_g ||b_g||^2``."""

    def __init__(
        self,
        groups=None,
        alpha=1.0,
        delta=(1.0,),
        group_weights=None,
        max_iter=3,
        eps=1e-6,
        tol=1e-10,
        update_function=None,
        standardize=False,
        fit_intercept=False,
        copy_X=True,
        warm_start=True,
        solver=None,
        solver_options=None,
    ):
        RidgedGroupLasso.__init__(
            self,
            groups=groups,
            alpha=alpha,
            delta=delta,
            group_weights=group_weights,
            standardize=standardize,
            fit_intercept=fit_intercept,
            copy_X=copy_X,
            warm_start=warm_start,
            solver=solver,
            solver_options=solver_options,
        )
        self._init_adaptive(max_iter, eps, tol, update_function)

    def _weights_to_penalty(self, weights, p, G):
        return np.zeros(p), weights, self._delta_vector(G)
