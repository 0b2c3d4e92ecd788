"""Standardised sparse-group penalty -- CPU oracle solver and optimality certificate.

TEST INFRASTRUCTURE (see oracle/__init__.py).  ``SparseGroupLasso(standardize=True)`` in the reference is

    1/(2n)||X b - y||^2 + sum_j a_j |b_j| + sum_g b_g ||X_g b_g||_2

(objective model/_lasso.py:109-121; regulariser :627-639 with the standardised group norms
``cp.norm2(X[:, mask] @ beta[mask])`` of :249-252; adaptive weights model/_adaptive_lasso.py:670-684).  The
group term has no closed-form proximal map next to the l1 term, so ``fista`` does not apply.  The oracle uses
the primal-dual iteration of Condat (2013) / Vu (2013) instead -- one dual block ``q_g`` per group with
``||q_g|| <= b_g``, ``sum_g b_g ||M_g b_g|| = max_q sum_g q_g^T M_g b_g`` and ``M_g^T M_g = X_g^T X_g``:

    b+  = soft(b - tau (grad f(b) + M^T q), tau a)
    q+  = project_g( q + sigma M (2 b+ - b) )            onto the balls ||q_g|| <= b_g

with ``tau (L_f / 2 + sigma ||M||^2) < 1``.  It shares nothing with the product's route (operator splitting
around weighted-l1 engine solves, sparselm_amd/model/_split.py): no inner solves, no augmented design.
``kkt_standardized`` checks a candidate against the optimality conditions of the ORIGINAL problem.
"""

from __future__ import annotations

import numpy as np


def group_factors(X, gidx, n_groups):
    """[(columns, M_g)] with M_g = S V^T of the thin SVD of X_g (rank(X_g) rows): ||M_g v|| = ||X_g v||."""
    out = []
    for g in range(n_groups):
        cols = np.flatnonzero(gidx == g)
        if not len(cols):
            out.append((cols, np.zeros((0, 0))))
            continue
        _, sv, vt = np.linalg.svd(X[:, cols], full_matrices=False)
        r = int(np.sum(sv > 1e-12 * max(sv[0], 1e-300)))
        out.append((cols, sv[:r, None] * vt[:r]))
    return out


def standardized_sparse_group(X, y, a, b, gidx, n_groups, beta0=None, tol=1e-13, max_iter=2_000_000, check_every=50):
    """Minimise 1/(2n)||X b - y||^2 + sum_j a_j|b_j| + sum_g b_g ||X_g b_g||_2.  Returns (beta, info).

    Stops when the optimality residual of ``kkt_standardized`` (evaluated every ``check_every`` iterations with
    the dual iterate as the certificate for inactive groups) is below ``tol`` times the scale of the gradient."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n, p = X.shape
    a = np.broadcast_to(np.asarray(a, dtype=np.float64), (p,))
    b = np.broadcast_to(np.asarray(b, dtype=np.float64), (n_groups,))
    fac = group_factors(X, gidx, n_groups)
    r = sum(M.shape[0] for _, M in fac)
    M = np.zeros((r, p))
    spans = []
    r0 = 0
    for cols, Mg in fac:
        M[r0 : r0 + Mg.shape[0], cols] = Mg
        spans.append((r0, r0 + Mg.shape[0]))
        r0 += Mg.shape[0]
    Lf = np.linalg.norm(X, 2) ** 2 / n if min(n, p) else 1.0
    Lm = np.linalg.norm(M, 2) if r else 0.0
    sigma = (Lf / max(Lm * Lm, 1e-300)) if Lm > 0 else 1.0  # balances the two terms of the step bound
    tau = 0.99 / (0.5 * Lf + sigma * Lm * Lm) if (Lf > 0 or Lm > 0) else 1.0
    beta = np.zeros(p) if beta0 is None else np.array(beta0, dtype=np.float64)
    q = np.zeros(r)
    Xty = X.T @ y / n
    gram = X.T @ X / n if p <= 4096 else None
    scale = max(np.max(np.abs(Xty)), 1e-300)
    it, resid, converged = 0, np.inf, False
    for it in range(1, max_iter + 1):
        grad = (gram @ beta if gram is not None else X.T @ (X @ beta) / n) - Xty
        v = beta - tau * (grad + M.T @ q)
        beta_new = np.sign(v) * np.maximum(np.abs(v) - tau * a, 0.0)
        qq = q + sigma * (M @ (2.0 * beta_new - beta))
        for g, (lo, hi) in enumerate(spans):
            nrm = np.linalg.norm(qq[lo:hi])
            if nrm > b[g]:
                qq[lo:hi] *= b[g] / nrm
        beta, q = beta_new, qq
        if it % check_every == 0:
            resid = kkt_standardized(X, y, a, b, gidx, n_groups, beta, factors=fac, dual=(q, spans))
            if resid <= tol * scale:
                converged = True
                break
    for g, (lo, hi) in enumerate(spans):  # groups the dual puts out are out exactly
        if hi > lo and np.linalg.norm(q[lo:hi]) < b[g] * (1.0 - 1e-12):
            beta[fac[g][0]] = 0.0
    return beta, {"n_iter": it, "converged": converged, "kkt": resid}


def kkt_standardized(X, y, a, b, gidx, n_groups, beta, factors=None, dual=None, zero_tol=0.0):
    """Largest violation of the optimality conditions of the standardised sparse-group problem at ``beta``:

    group with X_g b_g != 0:  h = grad_g + b_g X_g^T X_g b_g / ||X_g b_g||;  h_j + a_j sign(b_j) = 0 where
    b_j != 0, |h_j| <= a_j where b_j = 0;
    group with X_g b_g == 0:  some s (|s_j| <= a_j) and v (||v|| <= 1) with grad_g + s + b_g M_g^T v = 0.  For the
    second kind either ``dual`` = (q, spans) of the primal-dual iteration serves as the certificate (b_g v = q_g),
    or the best (s, v) is computed (``_inactive_violation``).  ``zero_tol``: coefficients up to this size count as
    zeros (an iterate a rounding away from a kink of the l1 term is judged by the interval condition there)."""
    X = np.asarray(X, dtype=np.float64)
    n, p = X.shape
    a = np.broadcast_to(np.asarray(a, dtype=np.float64), (p,))
    b = np.broadcast_to(np.asarray(b, dtype=np.float64), (n_groups,))
    fac = group_factors(X, gidx, n_groups) if factors is None else factors
    beta = np.where(np.abs(beta) <= zero_tol, 0.0, beta)
    grad = X.T @ (X @ beta - y) / n
    worst = 0.0
    for g, (cols, Mg) in enumerate(fac):
        if not len(cols):
            continue
        bg = beta[cols]
        fit = Mg @ bg
        nrm = np.linalg.norm(fit)
        if dual is not None:
            # with the dual iterate at hand a group is out when q_g lies strictly inside its ball (complementary
            # slackness: then M_g b_g = 0); without an l1 term the primal iterate only tends to zero there
            q, spans = dual
            lo, hi = spans[g]
            if np.linalg.norm(q[lo:hi]) < b[g] * (1.0 - 1e-12):
                res = grad[cols] + Mg.T @ q[lo:hi]
                worst = max(worst, float(np.max(np.maximum(np.abs(res) - a[cols], 0.0))))
                worst = max(worst, float(np.linalg.norm(Mg, 2) * nrm / n))  # what is left of the group, as a gradient
                continue
        if nrm > 0.0:
            h = grad[cols] + b[g] * (Mg.T @ fit) / nrm
            nz = bg != 0.0
            viol = np.where(nz, np.abs(h + a[cols] * np.sign(bg)), np.maximum(np.abs(h) - a[cols], 0.0))
            worst = max(worst, float(np.max(viol)))
            continue
        if np.any(bg != 0.0):  # X_g b_g = 0 with b_g != 0: only the l1 term sees these coordinates
            worst = max(worst, float(np.max(np.abs(bg))))
        if dual is not None:  # the dual iterate as the certificate (||q_g|| <= b_g by construction)
            res = grad[cols] + Mg.T @ dual[0][dual[1][g][0] : dual[1][g][1]]
            worst = max(worst, float(np.max(np.maximum(np.abs(res) - a[cols], 0.0))))
            continue
        worst = max(worst, _inactive_violation(grad[cols], a[cols], b[g], Mg))
    return worst


def _inactive_violation(grad_g, a_g, b_g, Mg, iters=20000):
    """min over |s_j| <= a_j, ||v|| <= 1 of || grad_g + s + b_g M_g^T v ||_inf-ish (returned as the max-norm of
    the residual at the minimiser of the 2-norm).  For fixed v the best s leaves soft(grad_g + b_g M_g^T v, a_g);
    phi(v) = 1/2 ||soft(.)||^2 is convex with gradient b_g M_g soft(.): accelerated projected gradient on the
    unit ball (dimension = rank of the group, a handful)."""
    soft = lambda t: np.sign(t) * np.maximum(np.abs(t) - a_g, 0.0)  # noqa: E731
    k = Mg.shape[0]
    if k == 0 or b_g == 0.0:
        return float(np.max(np.abs(soft(grad_g)))) if len(grad_g) else 0.0
    L = (b_g * np.linalg.norm(Mg, 2)) ** 2
    v = np.zeros(k)
    w, t = v.copy(), 1.0
    for _ in range(iters):
        res = soft(grad_g + b_g * (Mg.T @ w))
        if not np.any(res):
            v = w
            break
        v_new = w - (b_g * (Mg @ res)) / L
        nrm = np.linalg.norm(v_new)
        if nrm > 1.0:
            v_new /= nrm
        t_new = 0.5 * (1.0 + np.sqrt(1.0 + 4.0 * t * t))
        w = v_new + ((t - 1.0) / t_new) * (v_new - v)
        if np.linalg.norm(v_new - v) <= 1e-16 * max(1.0, np.linalg.norm(v_new)):
            v = v_new
            break
        v, t = v_new, t_new
    return float(np.max(np.abs(soft(grad_g + b_g * (Mg.T @ v)))))
