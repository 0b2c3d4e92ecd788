"""The solve seam between the estimator surface and the HIP engine.

``solve(problem)`` is the counterpart of ``CVXRegressor._solve`` (reference
src/sparselm/model/_base.py:512-519): it receives the already preprocessed ``(X, y)`` and the
penalty ``(a, b, d)`` on groups ``gidx`` and returns the minimiser.  The only backend shipped is
the HIP engine; ``use_backend`` exists so tests can time or check the surface against another
implementation without touching product code.
"""

from __future__ import annotations

import warnings
from contextlib import contextmanager

import numpy as np

from . import _engine

_KNOWN_OPTIONS = {"tol", "max_iter", "L", "restart", "check_every", "device"}


class SolveProblem:
    """Device-resident problem: upload once, solve many penalties (adaptive loops, paths)."""

    def __init__(self, backend, X, y, gidx, n_groups, options, row_weight=None, center=False):
        self.backend = backend
        self.options = options
        self.p = X.shape[1]
        self.n_groups = n_groups
        eng = _engine.get_engine(options.get("device"))
        self.ds = eng.dataset(X, y, row_weight=row_weight)
        # fit_intercept: centre the device copy in place (no centred host copy of X is ever made)
        self.x_mean, self.y_mean = self.ds.center() if center else (None, None)
        if gidx is not None:
            self.ds.set_groups(gidx, n_groups)

    def solve(self, a, b, d, beta0=None, want_group_norms=False):
        """One minimisation with penalty (a, b, d); returns (beta, group_norms or None, info)."""
        o = self.options
        flags = 0 if o.get("restart", True) else _engine.FLAG_NO_RESTART
        res = self.ds.solve_path(
            [(1.0, 1.0, 1.0)],
            a=a,
            b=b,
            d=d,
            beta0=beta0,
            tol=float(o.get("tol", default_tol(self.ds.n, self.ds.p))),
            max_iter=int(o.get("max_iter", 10000)),
            check_every=int(o.get("check_every", 0)),
            L=float(o.get("L", 0.0)),
            flags=flags,
            want_group_norms=want_group_norms,
        )
        if not res.converged:
            from sklearn.exceptions import ConvergenceWarning

            warnings.warn(
                f"FISTA did not reach tol={o.get('tol', default_tol(self.ds.n, self.ds.p)):g} in {int(res.n_iter[0])} iterations "
                f"(residual {res.resid[0]:.3e}); increase solver_options['max_iter'].",
                ConvergenceWarning,
            )
        info = {
            "n_iter": int(res.n_iter[0]),
            "converged": res.converged,
            "resid": float(res.resid[0]),
            "L": res.L,
            "loss": float(res.loss[0]),
            "wall_ms": res.wall_ms,
        }
        gn = None if res.group_norms is None else res.group_norms[0]
        return res.betas[0], gn, info

    def close(self):
        self.ds.close()


class HipBackend:
    name = "hip"
    # sample weights and centring are applied on the device (row weights in the fused kernel,
    # slm_dataset_center): the estimator hands over the raw validated arrays
    native_preprocessing = True

    def problem(self, X, y, gidx, n_groups, options, row_weight=None, center=False) -> SolveProblem:
        return SolveProblem(self, X, y, gidx, n_groups, options, row_weight=row_weight, center=center)


_backend = HipBackend()


def get_backend():
    return _backend


@contextmanager
def use_backend(backend):
    """Temporarily route solves through ``backend`` (test hook; the product never calls this)."""
    global _backend
    old = _backend
    _backend = backend
    try:
        yield backend
    finally:
        _backend = old


def default_tol(n: int, p: int) -> float:
    """Stopping tolerance when ``solver_options`` names none.  ``tol`` bounds the last prox step relative to
    ``||beta||``; the distance to the minimiser is about the condition number times that.  Small
    problems -- the reference's own sizes, often strongly correlated features -- are launch-bound, so two more
    digits cost little there: 1e-10 below 2^26 matrix entries, 1e-8 (1e-9...2e-8 from the minimiser on the
    BASELINE designs) above."""
    return 1e-10 if int(n) * int(p) < (1 << 26) else 1e-8


def normalise_options(solver_options) -> dict:
    """solver_options must be a dict (TypeError otherwise, as reference _base.py:198-199)."""
    if solver_options is None:
        return {}
    if not isinstance(solver_options, dict):
        raise TypeError("solver_options must be a dictionary")
    unknown = set(solver_options) - _KNOWN_OPTIONS
    if unknown:
        warnings.warn(
            f"solver_options {sorted(unknown)} are cvxpy/solver specific and are ignored by the HIP "
            f"engine (known: {sorted(_KNOWN_OPTIONS)})",
            UserWarning,
        )
    return {k: v for k, v in solver_options.items() if k in _KNOWN_OPTIONS}


def as_f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)
