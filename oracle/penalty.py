"""Penalty family of the sparse-lm Lasso estimators, restated for the CPU oracle.

TEST INFRASTRUCTURE (see oracle/__init__.py).  All arithmetic is IEEE fp64.

Every in-scope reference estimator minimises

    F(beta) = 1/(2n) ||X beta - y||^2                      (model/_lasso.py:109-121)
            + sum_j a_j |beta_j|                           (Lasso :99-107, SparseGroupLasso :627-639,
                                                            AdaptiveLasso model/_adaptive_lasso.py:167-175)
            + sum_g b_g ||beta_g||_2                       (GroupLasso :267-275, SGL :627-639,
                                                            AdaptiveGroupLasso _adaptive_lasso.py:354-362)
            + 1/2 sum_g d_g ||beta_g||_2^2                 (RidgedGroupLasso :795-811)

with ``a`` a per-coefficient vector, ``b`` and ``d`` per-group vectors; the penalties are not
divided by n (Appendix A-1 of SURVEY.md).
"""

from __future__ import annotations

import numpy as np


def group_index(groups, n_features: int):
    """Map arbitrary group labels to dense indices 0..G-1 in sorted-unique label order.

    Follows model/_lasso.py:248 (``group_masks = [groups == i for i in np.sort(np.unique(groups))]``):
    the i-th sorted label is group i, so ``group_weights[i]`` / ``delta[i]`` pair with it.
    ``groups=None`` means every feature is its own group (model/_lasso.py:211-217, :261).
    """
    if groups is None:
        return np.arange(n_features, dtype=np.int64), n_features
    labels = np.asarray(groups)
    uniq, inv = np.unique(labels, return_inverse=True)  # np.unique sorts
    return inv.astype(np.int64).reshape(-1), len(uniq)


def _group_norms(u, gidx, n_groups):
    return np.sqrt(np.bincount(gidx, weights=u * u, minlength=n_groups))


def prox(v, step, a, b, d, gidx, n_groups):
    """prox_{step * penalty}(v): soft-threshold, then block soft-threshold, then ridge shrink.

    ``u = sign(v) max(|v| - step a, 0)``; per group ``u_g * max(0, 1 - step b_g / ||u_g||) / (1 + step d_g)``.
    The composition is exact for the sparse-group penalty (l1 inside l2) and the ridge term only
    rescales the radial direction.
    """
    u = np.sign(v) * np.maximum(np.abs(v) - step * a, 0.0)
    nrm = _group_norms(u, gidx, n_groups)
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = np.where(nrm > 0.0, np.maximum(0.0, 1.0 - step * b / nrm), 0.0)
    scale = scale / (1.0 + step * d)
    return u * scale[gidx]


def penalty_value(beta, a, b, d, gidx, n_groups):
    nrm = _group_norms(beta, gidx, n_groups)
    return float(np.sum(a * np.abs(beta)) + np.sum(b * nrm) + 0.5 * np.sum(d * nrm * nrm))


def objective(X, y, beta, a, b, d, gidx, n_groups):
    """Value of F(beta); loss scaling 1/(2 n_samples) per model/_lasso.py:120."""
    r = X @ beta - y
    return float(r @ r) / (2.0 * X.shape[0]) + penalty_value(beta, a, b, d, gidx, n_groups)


def kkt_residual(grad, beta, a, b, d, gidx, n_groups):
    """Distance of 0 to the subdifferential of F at beta, given grad = X^T(X beta - y)/n.

    Returns the 2-norm of the per-coordinate / per-group violations:
      * group with beta_g != 0: h = grad_g + d_g beta_g + b_g beta_g/||beta_g||; for beta_j != 0 the
        violation is h_j + a_j sign(beta_j), for beta_j == 0 it is the distance of -h_j to [-a_j, a_j];
      * group with beta_g == 0: max(0, ||soft(grad_g, a_g)|| - b_g)  (existence of s in [-a,a], ||grad_g+s|| <= b_g).
    With mu = lambda_min(X^T X)/n > 0 this certifies ||beta - beta*||_2 <= kkt_residual / mu.
    """
    nrm = _group_norms(beta, gidx, n_groups)
    active_g = nrm > 0.0
    safe = np.where(active_g, nrm, 1.0)
    h = grad + (d[gidx] + b[gidx] / safe[gidx]) * beta
    viol = np.where(
        beta != 0.0,
        h + a * np.sign(beta),
        np.sign(h) * np.maximum(np.abs(h) - a, 0.0),
    )
    coord_part = np.where(active_g[gidx], viol, 0.0)
    s = np.sign(grad) * np.maximum(np.abs(grad) - a, 0.0)
    s = np.where(active_g[gidx], 0.0, s)
    zero_norm = _group_norms(s, gidx, n_groups)
    group_part = np.where(active_g, 0.0, np.maximum(0.0, zero_norm - b))
    return float(np.sqrt(np.sum(coord_part**2) + np.sum(group_part**2)))


def prox_fixed_point_residual(grad, beta, a, b, d, gidx, n_groups, step=1.0):
    """||beta - prox_step(beta - step grad)||_inf / step: zero iff beta is optimal."""
    return float(
        np.max(np.abs(beta - prox(beta - step * grad, step, a, b, d, gidx, n_groups))) / step
    )
