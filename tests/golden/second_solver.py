"""A second, independent solver for the penalty family -- test infrastructure, used only to PIN the oracle.

The oracle (oracle/fista.py) is a proximal-gradient iteration.  For the group / sparse-group / ridged /
adaptive fixtures no reference-held numbers exist and cvxpy is absent, so the oracle's solutions used to be
certified only by its own KKT residual.  This module computes the same minimisers a different way, sharing no
code with the oracle (no prox, no FISTA):

    active-set method  +  Newton's method on the optimality conditions of the active face.

On a face (a set A of non-zero coordinates with fixed signs, grouped into the active groups) the objective
    1/(2n) ||X b - y||^2 + sum_j a_j |b_j| + sum_g b_g ||b_g||_2 + 1/2 sum_g d_g ||b_g||_2^2
(reference src/sparselm/model/_lasso.py:109-121, 267-275, 627-639, 795-811) is smooth, and its stationarity
    X_A^T (X_A b_A - y) / n + a_A sign(b_A) + b_g b_g / ||b_g|| + d_g b_g = 0
is solved to machine precision with scipy.optimize.root.  The active set is grown from the empty one by the
worst violator of the optimality conditions (a group whose soft-thresholded gradient exceeds b_g, a coordinate
of an active group whose gradient exceeds a_j) and shrunk when the face solution flips a sign.  The final
point satisfies the conditions on EVERY coordinate; the objective is strictly convex on the fixtures (n > p), so
it is the unique minimiser.

The adaptive estimators are restated on top of it from the reference's formulas
(src/sparselm/model/_adaptive_lasso.py:158-232, 343-374, 654-726, 845-860).
"""

from __future__ import annotations

import numpy as np
from scipy.optimize import root


def _group_index(groups, p):
    if groups is None:
        return np.arange(p), p
    labels, gidx = np.unique(np.asarray(groups), return_inverse=True)  # sorted unique labels (_lasso.py:248)
    return gidx, len(labels)


def solve(X, y, a, b, d, gidx, G, max_rounds=500):
    """argmin of the objective above; a: (p,), b, d: (G,)."""
    X = np.asarray(X, float)
    y = np.asarray(y, float)
    n, p = X.shape
    a = np.broadcast_to(np.asarray(a, float), (p,))
    b = np.broadcast_to(np.asarray(b, float), (G,))
    d = np.broadcast_to(np.asarray(d, float), (G,))
    Gram = X.T @ X / n
    c = X.T @ y / n
    beta = np.zeros(p)
    sign = np.zeros(p)  # the face: sign[j] != 0 <=> j is active

    def face_solve():
        A = np.flatnonzero(sign)
        if not len(A):
            return
        gA = gidx[A]
        sA = sign[A]

        def F(v):
            out = Gram[np.ix_(A, A)] @ v - c[A] + a[A] * sA + d[gA] * v
            for g in np.unique(gA):
                m = gA == g
                nrm = np.linalg.norm(v[m])
                out[m] += b[g] * v[m] / max(nrm, 1e-300)
            return out

        start = beta[A].copy()
        start[start == 0] = 1e-3 * sA[start == 0]  # entrants leave zero on their side
        sol = root(F, start, method="hybr", tol=1e-15, options={"xtol": 1e-15, "maxfev": 20000})
        beta[:] = 0.0
        beta[A] = sol.x

    for _ in range(max_rounds):
        face_solve()
        # a coordinate that ended on the wrong side of zero leaves the face (then solve again)
        flipped = np.flatnonzero((sign != 0) & (beta * sign <= 0))
        if len(flipped):
            sign[flipped] = 0.0
            beta[flipped] = 0.0
            continue
        grad = Gram @ beta - c
        worst, worst_j = 0.0, None
        for g in range(G):
            idx = np.flatnonzero(gidx == g)
            if np.any(sign[idx]):  # active group: inactive members must satisfy |grad_j| <= a_j
                for j in idx[sign[idx] == 0]:
                    v = abs(grad[j]) - a[j]
                    if v > worst * 1.0 and v > 1e-13:
                        worst, worst_j = v, [j]
            else:  # inactive group: ||soft(grad_g, a_g)|| <= b_g
                s = np.sign(grad[idx]) * np.maximum(np.abs(grad[idx]) - a[idx], 0.0)
                v = np.linalg.norm(s) - b[g]
                if v > worst and v > 1e-13:
                    worst, worst_j = v, list(idx[s != 0])
        if worst_j is None:
            return beta
        for j in worst_j:
            sign[j] = -np.sign(grad[j])
    raise RuntimeError("active-set iteration did not settle")


def _center(X, y, fit_intercept):
    X = np.asarray(X, float)
    y = np.asarray(y, float)
    if not fit_intercept:
        return X, y, np.zeros(X.shape[1]), 0.0
    xm, ym = X.mean(axis=0), y.mean()
    return X - xm, y - ym, xm, ym


def group_lasso(X, y, groups, alpha, group_weights=None, l1_ratio=None, delta=None):
    """GroupLasso (l1_ratio None), SparseGroupLasso (l1_ratio given), RidgedGroupLasso (delta given)."""
    p = np.shape(X)[1]
    gidx, G = _group_index(groups, p)
    w = np.ones(G) if group_weights is None else np.asarray(group_weights, float)
    lam1, lam2 = (0.0, alpha) if l1_ratio is None else (l1_ratio * alpha, (1 - l1_ratio) * alpha)
    dd = np.zeros(G) if delta is None else np.broadcast_to(np.asarray(delta, float), (G,))
    return solve(X, y, lam1 * np.ones(p), lam2 * w, dd, gidx, G)


def adaptive(X, y, groups, alpha, group_weights=None, l1_ratio=None, delta=None, max_iter=3, eps=1e-6, tol=1e-10,
             fit_intercept=True):
    """AdaptiveGroupLasso / AdaptiveSparseGroupLasso / AdaptiveRidgedGroupLasso: first solve with alpha * ones
    (group_weights NOT applied, _adaptive_lasso.py:347-351, 658-667), then weights alpha * w_g * alpha /
    (||b_g|| + eps) (and lambda1 * alpha / (|b_j| + eps) for the l1 part, :721-726); stop on the change of the
    weight vector; the weights are updated after the last solve as well (:212-231)."""
    Xc, yc, xm, ym = _center(X, y, fit_intercept)
    p = Xc.shape[1]
    gidx, G = _group_index(groups, p)
    w = np.ones(G) if group_weights is None else np.asarray(group_weights, float)
    lam1, lam2 = (0.0, alpha) if l1_ratio is None else (l1_ratio * alpha, (1 - l1_ratio) * alpha)
    dd = np.zeros(G) if delta is None else np.broadcast_to(np.asarray(delta, float), (G,))
    a = lam1 * np.ones(p)
    b = lam2 * np.ones(G)
    prev = np.concatenate([b, a])
    n_iter = 0
    beta = None
    for _ in range(max_iter):
        beta = solve(Xc, yc, a, b, dd, gidx, G)
        n_iter += 1
        norms = np.sqrt(np.bincount(gidx, weights=beta * beta, minlength=G))
        a = lam1 * (alpha / (np.abs(beta) + eps))
        b = lam2 * w * (alpha / (norms + eps))
        new = np.concatenate([b, a])
        if np.linalg.norm(new - prev) <= tol:
            break
        prev = new
    return {"coef": beta, "intercept": ym - xm @ beta, "n_iter": n_iter, "group_weights": b, "l1_weights": a}
