"""Helpers that sit above ``fit``: box-constrained refits and an r2 -> CV-error conversion.

Counterparts of the reference's ``sparselm.tools`` (src/sparselm/tools.py:14-101 and :104-134):
pure host logic around any ``fit_method(X, y, ...) -> coefs`` callable, so they work unchanged
with the HIP-backed estimators.
"""

from __future__ import annotations

import functools
import warnings

import numpy as np


def _bound_vector(bound, count, default):
    """Scalar -> constant vector, None -> +-inf, array-like -> float array of length ``count``."""
    if bound is None:
        return np.full(count, default, dtype=float)
    if np.isscalar(bound):
        return np.full(count, float(bound))
    bound = np.asarray(bound, dtype=float)
    if bound.shape != (count,):
        raise ValueError(f"bounds must be scalars or have one entry per index ({count}), got {bound.shape}")
    return bound


def constrain_coefficients(indices, high=None, low=None):
    """Decorator factory: keep ``coefs[indices]`` of a fit method inside ``[low, high]``.

    Semantics follow reference tools.py:66-96: fit once; if any constrained coefficient leaves
    its interval, pin those coefficients at the violated bound, move their contribution to the
    right-hand side (``y -= X[:, j] * bound_j``), zero their columns, refit, and write the bounds
    into the result.  If the refit pushes *other* constrained coefficients out of range a
    ``RuntimeWarning`` is raised and the coefficients are returned as they are.

    Usage::

        coefs = constrain_coefficients(idx, high=2, low=0)(fit_method)(X, y)

        @constrain_coefficients(idx, high, low)
        def fit_method(X, y, ...): ...
    """
    idx = np.asarray(indices, dtype=np.intp).ravel()
    hi = _bound_vector(high, idx.size, np.inf)
    lo = _bound_vector(low, idx.size, -np.inf)

    def violations(coefs):
        picked = np.asarray(coefs)[idx]
        return picked > hi, picked < lo

    def decorator(fit_method):
        @functools.wraps(fit_method)
        def constrained_fit(X, y, *args, **kwargs):
            coefs = fit_method(X, y, *args, **kwargs)
            over, under = violations(coefs)
            if over.any() or under.any():
                Xc = np.array(X, dtype=float, copy=True)
                yc = np.array(y, dtype=float, copy=True)
                pinned = np.concatenate([idx[over], idx[under]])
                values = np.concatenate([hi[over], lo[under]])
                yc -= Xc[:, pinned] @ values
                Xc[:, pinned] = 0.0
                coefs = np.array(fit_method(Xc, yc, *args, **kwargs), copy=True)
                coefs[pinned] = values
                over, under = violations(coefs)
                if over.any() or under.any():
                    warnings.warn(
                        "The constrained refit moved other constrained coefficients out of their "
                        "bounds; check that the bounds are sensible for this problem.",
                        RuntimeWarning,
                    )
            return coefs

        return constrained_fit

    return decorator


def r2_score_to_cv_error(score, y, y_pred, weights=None):
    """Turn a cross-validated r2 score into a (weighted) RMS CV error (reference tools.py:104-134):
    ``sqrt((1 - score) * sum_i w_i (y_i - y_pred_i)^2 / sum_i w_i)``."""
    y = np.asarray(y, dtype=float)
    y_pred = np.asarray(y_pred, dtype=float)
    w = np.ones(y.shape[0]) if weights is None else np.asarray(weights, dtype=float)
    if w.shape[0] != y.shape[0]:
        raise ValueError("Weights given but not the same length as sample.")
    if (w < 0).any() or np.allclose(w, 0):
        raise ValueError("Weights can not be negative or all zero.")
    spread = np.sum(w * (y - y_pred) ** 2) / np.sum(w)
    return float(np.sqrt((1.0 - score) * spread))
