"""Accelerated proximal gradient (FISTA with gradient-scheme restart) -- CPU oracle solver.

TEST INFRASTRUCTURE (see oracle/__init__.py).  This stands in for the third-party call
``problem.solve(...)`` made at model/_base.py:516-518, model/_adaptive_lasso.py:213-215 and
model/_lasso.py:488-490: it returns the minimiser of the objective assembled at
model/_lasso.py:109-121.  The algorithm is not the reference's (that is whatever conic solver
cvxpy picks); only the minimiser is comparable, which is why the oracle runs to a much tighter
tolerance than the product and is itself certified by ``penalty.kkt_residual``.
"""

from __future__ import annotations

import numpy as np

from .penalty import prox


def lipschitz(X, exact_limit: int = 4_000_000, iters: int = 200, seed: int = 0) -> float:
    """lambda_max(X^T X)/n, the Lipschitz constant of the gradient of 1/(2n)||Xb-y||^2."""
    n, p = X.shape
    if n == 0 or p == 0:
        return 1.0
    if n * p <= exact_limit:
        s = np.linalg.norm(X, 2)
        return float(s * s) / n
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(p)
    v /= np.linalg.norm(v)
    lam = 0.0
    for _ in range(iters):
        w = X.T @ (X @ v)
        lam_new = float(np.linalg.norm(w))
        if lam_new == 0.0:
            return 1.0
        v = w / lam_new
        if abs(lam_new - lam) <= 1e-6 * lam_new:
            lam = lam_new
            break
        lam = lam_new
    return 1.02 * lam / n


def fista(
    X,
    y,
    a,
    b,
    d,
    gidx,
    n_groups,
    beta0=None,
    L=None,
    tol=1e-13,
    max_iter=200_000,
    restart=True,
):
    """Minimise 1/(2n)||X beta - y||^2 + pen_{a,b,d}(beta).  Returns (beta, info dict).

    Iteration: v = z - grad(z)/L; beta+ = prox_{1/L}(v); restart the momentum
    when (z - beta+)^T (beta+ - beta) > 0; t+ = (1+sqrt(1+4t^2))/2; z+ = beta+ + ((t-1)/t+)(beta+ - beta).
    Stops when ||beta+ - z||_2 <= tol * max(||beta+||_2, tiny)  (prox-gradient map residual at z).
    """
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n, p = X.shape
    a = np.broadcast_to(np.asarray(a, dtype=np.float64), (p,))
    b = np.broadcast_to(np.asarray(b, dtype=np.float64), (n_groups,))
    d = np.broadcast_to(np.asarray(d, dtype=np.float64), (n_groups,))
    if L is None:
        L = lipschitz(X)
    if not np.isfinite(L) or L <= 0.0:
        L = 1.0
    step = 1.0 / L
    beta = np.zeros(p) if beta0 is None else np.array(beta0, dtype=np.float64)
    z = beta.copy()
    t = 1.0
    n_restart = 0
    it = 0
    converged = False
    for it in range(1, max_iter + 1):
        g = X.T @ (X @ z - y) / n
        beta_new = prox(z - step * g, step, a, b, d, gidx, n_groups)
        resid = np.linalg.norm(beta_new - z)
        if restart and float((z - beta_new) @ (beta_new - beta)) > 0.0:
            t = 1.0
            n_restart += 1
        t_new = 0.5 * (1.0 + np.sqrt(1.0 + 4.0 * t * t))
        z = beta_new + ((t - 1.0) / t_new) * (beta_new - beta)
        beta = beta_new
        t = t_new
        if resid <= tol * max(np.linalg.norm(beta), np.finfo(float).tiny):
            converged = True
            break
    return beta, {"n_iter": it, "converged": converged, "restarts": n_restart, "L": L}
