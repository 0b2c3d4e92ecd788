"""``SparseGroupLasso(standardize=True)`` on the engine: an operator splitting around weighted-l1 solves.

The reference builds ``lambda1 ||b||_1 + lambda2 sum_g w_g ||X_g b_g||_2`` (src/sparselm/model/_lasso.py:616-639
with the standardised group norms of :249-252; the adaptive variant re-weights both terms,
_adaptive_lasso.py:670-684, 712-726) and leaves the rest to a conic solver.  No per-group change of variables
makes both terms separable at once -- the group norm is a plain l2 norm in ``gamma_g = M_g b_g`` (``M_g^T M_g =
X_g^T X_g``), the l1 norm only in ``b`` -- so the engine's proximal step does not cover the sum.  It covers each
half, though.  With ``gamma`` as a second block of unknowns tied to ``M b`` by a scaled multiplier ``u``
(alternating direction method of multipliers, over-relaxed):

    b      <- argmin 1/(2n)||X b - y||^2 + sum_j a_j |b_j| + rho/2 sum_g ||M_g b_g - gamma_g + u_g||^2
    gamma  <- group soft-threshold of (M b + u) at b_g / rho                      (closed form)
    u      <- u + M b - gamma

The first line is a weighted Lasso on the design ``[X; sqrt(n rho) M]`` whose last rows carry the targets
``sqrt(n rho) (gamma - u)``: ONE upload, then per sweep a new target vector (``slm_dataset_set_targets``) and a
warm-started engine solve.  ``rho`` starts at 1/n (the coupling term then has the curvature of the loss) and is
re-balanced a few times from the primal / dual residuals (each change rebuilds the extra rows).  The sweeps stop
when both residuals are below ``tol`` relative to the quantities they compare (Boyd et al. 2011, section 3.3).
"""

from __future__ import annotations

import warnings

import numpy as np

from .._backend import default_tol, get_backend

_REBALANCE_AT = (5, 10, 20, 40, 80, 160, 320)
_RELAX = 1.6
_MAX_SWEEPS = 5000


class StandardizedSparseGroupProblem:
    """Looks like a backend problem (``solve(a, b, d, beta0, want_group_norms)``, ``close()``) to the estimators;
    every ``solve`` is a run of the splitting above.  ``group_norms`` are ``||X_g b_g||_2`` -- what the reference's
    adaptive update reads (_adaptive_lasso.py:712-726 through ``auxiliaries.group_norms.value``)."""

    def __init__(self, X, y, gidx, n_groups, options):
        X = np.asarray(X, dtype=np.float64)
        self.X, self.y = X, np.asarray(y, dtype=np.float64)
        self.n, self.p = X.shape
        self.G = int(n_groups)
        self.options = dict(options)
        self.gidx = np.arange(self.p) if gidx is None else np.asarray(gidx)
        self.inner = None
        self.blocks = None
        # Problems the on-chip solver takes run ALL sweeps in one launch on the dataset of (X, y) itself
        # (slm_solve_standardized_sgl, csrc/small_split_kernels.hpp); the sweeps below are the general route, and
        # the A/B partner of that kernel (option ``on_chip=False``).
        self.dev = None
        self.dev_warm = False
        # (only where that kernel can take the problem -- p <= 128 and n * ld <= 2^17, the rule of slm_solve_standardized_sgl:
        #  for anything larger a device dataset opened here would be an upload of X, learned to be useless at the first solve)
        ld = (self.p + 15) // 16 * 16
        if self.options.get("on_chip", True) is not False and self.p <= 128 and self.n * ld <= 131072:
            self.dev = get_backend().problem(self.X, self.y, gidx, self.G, self.options)
            if not hasattr(getattr(self.dev, "ds", None), "solve_standardized_sgl"):  # (the tests' CPU stand-in)
                self.dev.close()
                self.dev = None
        if self.dev is None:
            self._host_setup()

    def _host_setup(self):
        X, gidx = self.X, self.gidx
        # M_g = S_g V_g^T of the thin SVD of X_g (rank-deficient groups keep rank(X_g) rows)
        blocks, r0 = [], 0
        for g in range(self.G):
            cols = np.flatnonzero(gidx == g)
            if not len(cols):
                blocks.append((cols, np.zeros((0, 0)), r0, r0))
                continue
            _, sv, vt = np.linalg.svd(X[:, cols], full_matrices=False)
            r = int(np.sum(sv > 1e-12 * max(sv[0], 1e-300))) if len(sv) else 0
            blocks.append((cols, sv[:r, None] * vt[:r], r0, r0 + r))
            r0 += r
        self.blocks, self.r = blocks, r0
        self.M = np.zeros((self.r, self.p))
        for cols, Mg, lo, hi in blocks:
            self.M[lo:hi, cols] = Mg
        self.scale = np.sqrt((self.n + self.r) / self.n)  # the engine's loss is 1/(2 rows)
        self.rho = 1.0 / self.n
        self.gamma = np.zeros(self.r)
        self.u = np.zeros(self.r)
        self._build()

    def _build(self):
        if self.inner is not None:
            self.inner.close()
            self.inner = None
        Xa = self.scale * np.vstack([self.X, np.sqrt(self.n * self.rho) * self.M])
        ya = self.scale * np.concatenate([self.y, np.zeros(self.r)])
        inner_options = dict(self.options)
        inner_options.setdefault("tol", min(default_tol(self.n, self.p), 1e-10))
        # (not through the dataset cache: the targets of this dataset change under it)
        self.inner = get_backend().problem(Xa, ya, None, self.p, inner_options, cache=False)

    def _targets(self):
        tail = np.sqrt(self.n * self.rho) * (self.gamma - self.u)
        return self.scale * np.concatenate([self.y, tail])

    def _group_shrink(self, v, b):
        out = np.array(v)
        for g, (_, _, lo, hi) in enumerate(self.blocks):
            if hi == lo:
                continue
            nrm = np.linalg.norm(v[lo:hi])
            thr = b[g] / self.rho
            out[lo:hi] = 0.0 if nrm <= thr else v[lo:hi] * (1.0 - thr / nrm)
        return out

    def group_norms(self, beta):
        v = self.M @ beta
        return np.array([np.linalg.norm(v[lo:hi]) for _, _, lo, hi in self.blocks])

    def _solve_on_chip(self, a, b, beta0, want_group_norms, tol):
        """All sweeps in one launch; ``None`` when the problem is not one the kernel takes, or it ran out of sweeps."""
        o = self.options
        try:
            beta, gn, rec = self.dev.ds.solve_standardized_sgl(
                a, b, beta0=beta0, warm=self.dev_warm, tol=tol, tol_inner=float(o["tol"]) if "tol" in o else min(tol, 1e-10),
                max_sweeps=_MAX_SWEEPS, want_group_norms=want_group_norms,
            )
        except NotImplementedError:
            return None
        self.dev_warm = True
        if int(rec["status"]) != 0:
            return None
        info = {"n_iter": int(rec["n_iter"]), "converged": True, "resid": float(rec["resid"]),
                "inner_iterations": int(rec["rejects"]), "rho": float(rec["L"]), "on_chip": True}
        return beta, gn, info

    def solve(self, a, b, d, beta0=None, want_group_norms=False):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        if d is not None and np.any(np.asarray(d) != 0.0):
            raise ValueError("the splitting for standardised sparse-group penalties carries no ridge term")
        tol = float(self.options.get("tol", default_tol(self.n, self.p)))
        if self.dev is not None:
            done = self._solve_on_chip(a, b, beta0, want_group_norms, tol)
            if done is not None:
                return done
            self.dev.close()  # (not a problem for the kernel: the sweeps below take this and every later call)
            self.dev = None
        if self.blocks is None:
            self._host_setup()
        zeros_g = np.zeros(self.p)  # (the inner problem has singleton groups: one entry per feature)
        beta = None if beta0 is None else np.asarray(beta0, dtype=np.float64)
        if beta is not None and not np.any(self.gamma) and not np.any(self.u):
            self.gamma = self.M @ beta
        inner_iters = 0
        converged = False
        rp = rd = np.inf
        sweeps = 0
        for sweeps in range(1, _MAX_SWEEPS + 1):
            self.inner.set_targets(self._targets())
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")  # an inner solve short of its tolerance is absorbed by the sweeps
                beta, _, info = self.inner.solve(a, zeros_g, zeros_g, beta0=beta)
            inner_iters += int(info.get("n_iter", 0))
            v = self.M @ beta
            vh = _RELAX * v + (1.0 - _RELAX) * self.gamma
            gamma_new = self._group_shrink(vh + self.u, b)
            self.u = self.u + vh - gamma_new
            rp = np.linalg.norm(v - gamma_new)
            rd = self.rho * np.linalg.norm(self.M.T @ (gamma_new - self.gamma))
            self.gamma = gamma_new
            ep = max(np.linalg.norm(v), np.linalg.norm(self.gamma), 1e-300)
            ed = max(self.rho * np.linalg.norm(self.M.T @ self.u), 1e-300)
            if rp <= tol * ep and rd <= tol * ed:
                converged = True
                break
            if sweeps in _REBALANCE_AT:
                ratio = (rp / ep) / max(rd / ed, 1e-300)
                if ratio > 5.0 or ratio < 0.2:
                    factor = min(10.0, max(0.1, np.sqrt(ratio)))
                    self.u = self.u / factor  # u is the multiplier divided by rho
                    self.rho *= factor
                    self._build()
        # gamma is exactly group-sparse, M b only to the residual: a group whose gamma_g vanished is out
        for (cols, _, lo, hi) in self.blocks:
            if hi > lo and not np.any(self.gamma[lo:hi]):
                beta[cols] = 0.0
        if not converged:
            from sklearn.exceptions import ConvergenceWarning

            warnings.warn(
                f"the splitting for the standardised sparse-group penalty did not reach tol={tol:g} in {sweeps} sweeps "
                f"(primal residual {rp:.3e}, dual residual {rd:.3e})",
                ConvergenceWarning,
            )
        info = {"n_iter": sweeps, "converged": converged, "resid": float(max(rp, rd)), "inner_iterations": inner_iters,
                "rho": self.rho}
        return beta, (self.group_norms(beta) if want_group_norms else None), info

    def close(self):
        if self.dev is not None:
            self.dev.close()
            self.dev = None
        if self.inner is not None:
            self.inner.close()
            self.inner = None
