"""Estimator-level semantics of the reference, restated on top of the oracle solver.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Plain functions, one per in-scope reference
estimator; each returns a dict with ``coef``, ``intercept`` and, for Adaptive*, ``n_iter`` and the
final ``weights``.  Citations are relative to /root/reference/src/sparselm.
"""

from __future__ import annotations

import numpy as np

from .fista import fista
from .penalty import group_index
from .primal_dual import group_factors, standardized_sparse_group


# --------------------------------------------------------------------------------------------
# preprocessing: model/_base.py:207-227 + sklearn.linear_model._base._preprocess_data /
# _rescale_data / LinearModel._set_intercept (restated; X_scale is identically 1)
# --------------------------------------------------------------------------------------------
def preprocess(X, y, sample_weight=None, fit_intercept=False):
    X = np.array(X, dtype=np.float64)
    y = np.array(y, dtype=np.float64)
    n = X.shape[0]
    sw = None
    if sample_weight is not None:
        sw = np.broadcast_to(np.asarray(sample_weight, dtype=np.float64), (n,)).copy()
        sw = sw * (n / np.sum(sw))  # _base.py:214  rescale to sum to n_samples
    if fit_intercept:
        X_offset = np.average(X, axis=0, weights=sw)
        y_offset = float(np.average(y, weights=sw))
        X = X - X_offset
        y = y - y_offset
    else:
        X_offset = np.zeros(X.shape[1])
        y_offset = 0.0
    if sw is not None:  # _base.py:224-225 -> rows scaled by sqrt(w)
        s = np.sqrt(sw)
        X = X * s[:, None]
        y = y * s
    return X, y, X_offset, y_offset


def _intercept(coef, X_offset, y_offset, fit_intercept):
    # LinearModel._set_intercept: intercept_ = y_offset - X_offset @ coef_, else exactly 0.0
    return float(y_offset - X_offset @ coef) if fit_intercept else 0.0


def _solve(X, y, a, b, d, gidx, G, beta0=None, **kw):
    beta, info = fista(X, y, a, b, d, gidx, G, beta0=beta0, **kw)
    return beta, info


def _group_weights(group_weights, G):
    # model/_lasso.py:233-235: default group weights are ones(G) (not sqrt(size))
    return np.ones(G) if group_weights is None else np.asarray(group_weights, dtype=np.float64)


def _whiten_groups(X, gidx, G):
    """standardize=True (model/_lasso.py:249-252 penalises ||X_g beta_g||): per-group SVD
    X_g = U S V^T, gamma_g = S V^T beta_g  =>  ||X_g beta_g|| = ||gamma_g|| on the design U.
    (The product uses QR; a different factorisation here keeps the cross-check independent.)"""
    U = np.empty_like(X)
    maps = []
    for g in range(G):
        cols = np.flatnonzero(gidx == g)
        u, sv, vt = np.linalg.svd(X[:, cols], full_matrices=False)
        U[:, cols] = u
        maps.append((cols, vt.T / sv))
    def back(gamma):
        beta = np.empty_like(gamma)
        for cols, m in maps:
            beta[cols] = m @ gamma[cols]
        return beta
    return U, back


def _delta(delta, G):
    # model/_lasso.py:744-765: delta of length 1 is broadcast to all groups
    delta = np.asarray(delta, dtype=np.float64).reshape(-1)
    return delta * np.ones(G) if len(delta) == 1 else delta


# --------------------------------------------------------------------------------------------
# non-adaptive estimators
# --------------------------------------------------------------------------------------------
def fit_lasso(X, y, alpha=1.0, fit_intercept=False, sample_weight=None, **kw):
    """Lasso: model/_lasso.py:34-121 (objective :109-121, regulariser :99-107)."""
    Xp, yp, xo, yo = preprocess(X, y, sample_weight, fit_intercept)
    p = Xp.shape[1]
    gidx, G = group_index(None, p)
    beta, info = _solve(Xp, yp, alpha * np.ones(p), np.zeros(G), np.zeros(G), gidx, G, **kw)
    return {"coef": beta, "intercept": _intercept(beta, xo, yo, fit_intercept), "info": info}


def fit_group_lasso(
    X, y, groups=None, alpha=1.0, group_weights=None, fit_intercept=False, sample_weight=None,
    standardize=False, **kw
):
    """GroupLasso: model/_lasso.py:124-275 (group norms :239-255, regulariser :267-275)."""
    Xp, yp, xo, yo = preprocess(X, y, sample_weight, fit_intercept)
    p = Xp.shape[1]
    gidx, G = group_index(groups, p)
    w = _group_weights(group_weights, G)
    back = None
    if standardize:
        Xp, back = _whiten_groups(Xp, gidx, G)
    beta, info = _solve(Xp, yp, np.zeros(p), alpha * w, np.zeros(G), gidx, G, **kw)
    if back is not None:
        beta = back(beta)
    return {"coef": beta, "intercept": _intercept(beta, xo, yo, fit_intercept), "info": info}


def fit_sparse_group_lasso(
    X,
    y,
    groups=None,
    l1_ratio=0.5,
    alpha=1.0,
    group_weights=None,
    fit_intercept=False,
    sample_weight=None,
    standardize=False,
    **kw,
):
    """SparseGroupLasso: model/_lasso.py:505-639; lambda1 = l1_ratio*alpha, lambda2 = (1-l1_ratio)*alpha (:616-625).
    ``standardize``: the group norms are ||X_g beta_g|| (:249-252), X the preprocessed design."""
    Xp, yp, xo, yo = preprocess(X, y, sample_weight, fit_intercept)
    p = Xp.shape[1]
    gidx, G = group_index(groups, p)
    w = _group_weights(group_weights, G)
    lam1, lam2 = l1_ratio * alpha, (1.0 - l1_ratio) * alpha
    if standardize:
        beta, info = standardized_sparse_group(Xp, yp, lam1 * np.ones(p), lam2 * w, gidx, G)
        return {"coef": beta, "intercept": _intercept(beta, xo, yo, fit_intercept), "info": info}
    beta, info = _solve(Xp, yp, lam1 * np.ones(p), lam2 * w, np.zeros(G), gidx, G, **kw)
    return {"coef": beta, "intercept": _intercept(beta, xo, yo, fit_intercept), "info": info}


def fit_ridged_group_lasso(
    X,
    y,
    groups=None,
    alpha=1.0,
    delta=(1.0,),
    group_weights=None,
    fit_intercept=False,
    sample_weight=None,
    **kw,
):
    """RidgedGroupLasso (standardize=False): model/_lasso.py:642-811; + 0.5*sum_g delta_g ||beta_g||^2 (:795-811)."""
    Xp, yp, xo, yo = preprocess(X, y, sample_weight, fit_intercept)
    p = Xp.shape[1]
    gidx, G = group_index(groups, p)
    w = _group_weights(group_weights, G)
    beta, info = _solve(Xp, yp, np.zeros(p), alpha * w, _delta(delta, G), gidx, G, **kw)
    return {"coef": beta, "intercept": _intercept(beta, xo, yo, fit_intercept), "info": info}


def overlap_extension(group_list, n_features):
    """Column duplication that makes overlapping groups disjoint (model/_lasso.py:440-461):
    group ids sorted; for every group the features that list it; returns (beta_indices,
    extended_groups).  ``group_list=None`` means singleton groups (:446-448)."""
    if group_list is None:
        group_list = [[i] for i in range(n_features)]
    group_ids = np.sort(np.unique([gid for grp in group_list for gid in grp]))
    inds = [[i for i, grp in enumerate(group_list) if gid in grp] for gid in group_ids]
    extended_groups = np.concatenate([len(g) * [i] for i, g in enumerate(inds)])
    return np.concatenate(inds).astype(np.int64), extended_groups.astype(np.int64)


def fit_overlap_group_lasso(
    X, y, group_list=None, alpha=1.0, group_weights=None, fit_intercept=False, sample_weight=None, **kw
):
    """OverlapGroupLasso: group lasso on the column-duplicated design, coefficients of duplicated
    columns summed back (model/_lasso.py:279-502, fold-back :486-502)."""
    Xp, yp, xo, yo = preprocess(X, y, sample_weight, fit_intercept)
    p = Xp.shape[1]
    bidx, ext = overlap_extension(group_list, p)
    G = int(ext.max()) + 1
    w = _group_weights(group_weights, G)
    beta_ext, info = _solve(Xp[:, bidx], yp, np.zeros(len(bidx)), alpha * w, np.zeros(G), ext, G, **kw)
    beta = np.bincount(bidx, weights=beta_ext, minlength=p)
    return {"coef": beta, "intercept": _intercept(beta, xo, yo, fit_intercept), "info": info}


def fit_adaptive_overlap_group_lasso(
    X, y, group_list=None, alpha=1.0, group_weights=None, max_iter=3, eps=1e-6, tol=1e-10,
    update_function=None, fit_intercept=False, sample_weight=None, **kw
):
    """AdaptiveOverlapGroupLasso: AdaptiveGroupLasso on the extended design, folded back
    (model/_adaptive_lasso.py:377-524)."""
    Xp, yp, xo, yo = preprocess(X, y, sample_weight, fit_intercept)
    p = Xp.shape[1]
    bidx, ext = overlap_extension(group_list, p)
    r = fit_adaptive_group_lasso(
        Xp[:, bidx], yp, groups=ext, alpha=alpha, group_weights=group_weights, max_iter=max_iter, eps=eps,
        tol=tol, update_function=update_function, fit_intercept=False, **kw
    )
    beta = np.bincount(bidx, weights=r["coef"], minlength=p)
    return {"coef": beta, "intercept": _intercept(beta, xo, yo, fit_intercept), "n_iter": r["n_iter"],
            "weights": r["weights"]}


# --------------------------------------------------------------------------------------------
# adaptive (iteratively re-weighted) estimators: model/_adaptive_lasso.py
# --------------------------------------------------------------------------------------------
def _default_update(alpha):
    # model/_adaptive_lasso.py:177-182: update(beta, eps) = alpha / (|beta| + eps)
    return lambda x, eps: alpha / (np.abs(x) + eps)


def _adaptive_loop(solve_with, update_weights, weights0, max_iter, tol):
    """Outer loop of AdaptiveLasso._solve, model/_adaptive_lasso.py:206-232.

    ``solve_with(weights, beta_prev)`` -> beta;  ``update_weights(beta)`` -> new flat weight vector.
    Weights are updated after EVERY solve including the last (:223-227); the loop stops early when
    ||w_new - w_prev||_2 <= tol (:189-194, :229-230); ``n_iter`` counts solves (:222); the returned
    beta is that of the last solve (:232).
    """
    weights = weights0
    previous = weights0.copy()
    beta = None
    n_iter = 0
    for i in range(max_iter):
        beta = solve_with(weights, beta)
        n_iter = i + 1
        weights = update_weights(beta)
        if np.linalg.norm(weights - previous) <= tol:
            break
        previous = weights.copy()
    return beta, n_iter, weights


def fit_adaptive_lasso(
    X,
    y,
    alpha=1.0,
    max_iter=3,
    eps=1e-6,
    tol=1e-10,
    update_function=None,
    fit_intercept=False,
    sample_weight=None,
    **kw,
):
    """AdaptiveLasso: weights start at alpha*1 (:158-165), then alpha*update(beta, eps) (:196-204),
    i.e. alpha^2/(|beta|+eps) with the default update (:177-182)."""
    Xp, yp, xo, yo = preprocess(X, y, sample_weight, fit_intercept)
    p = Xp.shape[1]
    gidx, G = group_index(None, p)
    update = _default_update(alpha) if update_function is None else update_function
    zeros = np.zeros(G)

    def solve_with(w, beta_prev):
        return _solve(Xp, yp, w, zeros, zeros, gidx, G, beta0=beta_prev, **kw)[0]

    def update_weights(beta):
        return alpha * np.asarray(update(beta, eps), dtype=np.float64)

    if max_iter == 0:
        raise ValueError("oracle: max_iter=0 returns beta.value=None in the reference")
    beta, n_iter, w = _adaptive_loop(solve_with, update_weights, alpha * np.ones(p), max_iter, tol)
    return {
        "coef": beta,
        "intercept": _intercept(beta, xo, yo, fit_intercept),
        "n_iter": n_iter,
        "weights": w,
    }


def fit_adaptive_group_lasso(
    X,
    y,
    groups=None,
    alpha=1.0,
    group_weights=None,
    max_iter=3,
    eps=1e-6,
    tol=1e-10,
    update_function=None,
    fit_intercept=False,
    sample_weight=None,
    delta=None,
    standardize=False,
    **kw,
):
    """AdaptiveGroupLasso: first solve uses alpha*ones(G) WITHOUT group_weights (:343-352, :354-362);
    later b_g = (alpha*w_g)*update(||beta_g||, eps) (:364-374).  ``delta`` != None gives
    AdaptiveRidgedGroupLasso (:729-860, ridge term model/_lasso.py:795-811)."""
    Xp, yp, xo, yo = preprocess(X, y, sample_weight, fit_intercept)
    p = Xp.shape[1]
    gidx, G = group_index(groups, p)
    gw = _group_weights(group_weights, G)
    dvec = np.zeros(G) if delta is None else _delta(delta, G)
    update = _default_update(alpha) if update_function is None else update_function
    zeros_p = np.zeros(p)
    back = None
    if standardize:  # group_norms.value is ||X_g beta_g|| then (:372-374): loop in whitened coordinates
        Xp, back = _whiten_groups(Xp, gidx, G)

    def solve_with(w, beta_prev):
        return _solve(Xp, yp, zeros_p, w, dvec, gidx, G, beta0=beta_prev, **kw)[0]

    def update_weights(beta):
        norms = np.sqrt(np.bincount(gidx, weights=beta * beta, minlength=G))
        return (alpha * gw) * np.asarray(update(norms, eps), dtype=np.float64)

    beta, n_iter, w = _adaptive_loop(solve_with, update_weights, alpha * np.ones(G), max_iter, tol)
    if back is not None:
        beta = back(beta)
    return {
        "coef": beta,
        "intercept": _intercept(beta, xo, yo, fit_intercept),
        "n_iter": n_iter,
        "weights": w,
    }


def fit_adaptive_ridged_group_lasso(X, y, groups=None, alpha=1.0, delta=(1.0,), **kw):
    """AdaptiveRidgedGroupLasso (standardize=False): model/_adaptive_lasso.py:729-860."""
    return fit_adaptive_group_lasso(X, y, groups=groups, alpha=alpha, delta=delta, **kw)


def fit_adaptive_sparse_group_lasso(
    X,
    y,
    groups=None,
    l1_ratio=0.5,
    alpha=1.0,
    group_weights=None,
    max_iter=3,
    eps=1e-6,
    tol=1e-10,
    update_function=None,
    fit_intercept=False,
    sample_weight=None,
    standardize=False,
    **kw,
):
    """AdaptiveSparseGroupLasso: a0 = lambda1*1, b0 = lambda2*1 (:654-668); updates
    a = lambda1*update(beta), b = (lambda2*w_g)*update(||beta_g||) (:712-726); convergence is checked
    on the concatenation [b, a] (:698-710).  ``standardize``: penalised and fed to the update are
    ||X_g beta_g|| (``auxiliaries.group_norms.value`` with model/_lasso.py:249-252)."""
    Xp, yp, xo, yo = preprocess(X, y, sample_weight, fit_intercept)
    p = Xp.shape[1]
    gidx, G = group_index(groups, p)
    gw = _group_weights(group_weights, G)
    lam1, lam2 = l1_ratio * alpha, (1.0 - l1_ratio) * alpha
    update = _default_update(alpha) if update_function is None else update_function
    zeros_g = np.zeros(G)

    fac = group_factors(Xp, gidx, G) if standardize else None

    def solve_with(w, beta_prev):
        if standardize:
            return standardized_sparse_group(Xp, yp, w[G:], w[:G], gidx, G, beta0=beta_prev)[0]
        return _solve(Xp, yp, w[G:], w[:G], zeros_g, gidx, G, beta0=beta_prev, **kw)[0]

    def update_weights(beta):
        if standardize:
            norms = np.array([np.linalg.norm(Mg @ beta[cols]) for cols, Mg in fac])
        else:
            norms = np.sqrt(np.bincount(gidx, weights=beta * beta, minlength=G))
        a_new = lam1 * np.asarray(update(beta, eps), dtype=np.float64)
        b_new = (lam2 * gw) * np.asarray(update(norms, eps), dtype=np.float64)
        return np.concatenate((b_new, a_new))

    w0 = np.concatenate((lam2 * np.ones(G), lam1 * np.ones(p)))
    beta, n_iter, w = _adaptive_loop(solve_with, update_weights, w0, max_iter, tol)
    return {
        "coef": beta,
        "intercept": _intercept(beta, xo, yo, fit_intercept),
        "n_iter": n_iter,
        "weights": w,
    }
