#!/usr/bin/env python3
"""Does a device-to-device copy of a dataset give the numbers of the original, bit for bit?  The grid search of
tests/test_model_selection.py::test_fast_path_batches_dealt_to_several_streams_give_the_same_search, run (a) on
the dataset itself, (b) with every batch on a copy, (c) solves on the copy and scoring on the original,
(d) solves on the original and scoring on the copy."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "sparse-lm_amd"))
from sklearn.model_selection import KFold  # noqa: E402

from sparselm_amd.model import SparseGroupLasso  # noqa: E402
from sparselm_amd.model_selection import GridSearchCV  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "lasso_family_golden.npz"))
X, y, groups = g["grp_X"], g["grp_y"], g["grp_groups"]
cv = KFold(5, shuffle=True, random_state=0)
grid = {"alpha": list(np.geomspace(10, 0.1, 5)), "l1_ratio": [0.1, 0.5, 0.9]}
est = SparseGroupLasso(groups=groups, fit_intercept=True, solver_options={"tol": 1e-11})


class Mixed:
    """solves on one dataset, scoring on another"""

    def __init__(self, solve_ds, score_ds):
        self._solve, self._score = solve_ds, score_ds

    def eval_sse(self, *a, **k):
        return self._score.eval_sse(*a, **k)

    def __getattr__(self, name):
        return getattr(self._solve, name)


def variant(which):
    def run(self, ds, batches, run_batch, grid):
        c = ds.clone()
        if grid.gidx is not None:
            grid._set_groups(c)
        try:
            target = {"copy": c, "solve_on_copy": Mixed(c, ds), "score_on_copy": Mixed(ds, c)}[which]
            return sum(run_batch(target, b) for b in batches)
        finally:
            eng = c.engine
            c.close()
            eng.close()

    return run


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    base = GridSearchCV(est, grid, cv=cv, lanes=4, streams=1).fit(X, y)
    plain = GridSearchCV._run_batches
    for which in ("copy", "solve_on_copy", "score_on_copy"):
        GridSearchCV._run_batches = variant(which)
        other = GridSearchCV(est, grid, cv=cv, lanes=4, streams=1).fit(X, y)
        GridSearchCV._run_batches = plain
        worst = 0.0
        for f in range(5):
            a, b = base.cv_results_[f"split{f}_test_score"], other.cv_results_[f"split{f}_test_score"]
            worst = max(worst, float(np.max(np.abs(a - b) / np.abs(a))))
        print(f"{which}: worst relative difference of a fold score {worst:.3e}", flush=True)
    for rep in range(3):
        again = GridSearchCV(est, grid, cv=cv, lanes=4, streams=1).fit(X, y)
        worst = max(float(np.max(np.abs(base.cv_results_[f"split{f}_test_score"] - again.cv_results_[f"split{f}_test_score"])))
                    for f in range(5))
        print(f"the original again: worst absolute difference {worst:.3e}")
