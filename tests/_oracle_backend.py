"""Backend that routes the estimator surface through the CPU oracle -- for CPU-only surface tests.

Lives under tests/ on purpose: the product package never imports the oracle.
"""

import numpy as np

import oracle


class _OracleProblem:
    def __init__(self, X, y, gidx, n_groups, options):
        self.X, self.y = np.asarray(X, float), np.asarray(y, float)
        p = self.X.shape[1]
        self.gidx = np.arange(p) if gidx is None else np.asarray(gidx, dtype=np.int64)
        self.G = n_groups
        self.options = options
        self.L = oracle.lipschitz(self.X)

    def solve(self, a, b, d, beta0=None, want_group_norms=False):
        beta, info = oracle.fista(
            self.X, self.y, a, b, d, self.gidx, self.G, beta0=beta0, L=self.L,
            tol=float(self.options.get("tol", 1e-12)), max_iter=int(self.options.get("max_iter", 200000)),
        )
        gn = None
        if want_group_norms:
            gn = np.sqrt(np.bincount(self.gidx, weights=beta * beta, minlength=self.G))
        return beta, gn, info

    def set_targets(self, y):
        self.y = np.asarray(y, float)

    def close(self):
        pass


class OracleBackend:
    name = "oracle"

    def problem(self, X, y, gidx, n_groups, options, cache=True):
        return _OracleProblem(X, y, gidx, n_groups, options)
