"""Grid search with the one-standard-error rule, grid-aware on the GPU.

Counterpart of ``sparselm.model_selection.GridSearchCV`` (reference
src/sparselm/model_selection.py:30-424): same constructor (``opt_selection_method`` in
{"max_score", "one_std_score"}, default scoring ``neg_root_mean_squared_error`` :166), same
selection rule (:190-223), same fitted attributes (``cv_results_``, ``best_index_``,
``best_params_``, ``best_score_``, ``best_score_std_`` :396-398, ``best_estimator_``).

What differs is how the (candidate x fold) grid is evaluated.  The reference dispatches
``clone(estimator).fit(X[train], y[train])`` per cell through joblib (:273, :304-323), re-indexing X
every time.  Here, for the Lasso-family estimators with a grid over ``alpha`` (and any other penalty
hyper-parameter, e.g. ``l1_ratio``), X is uploaded ONCE and

  * every fold is a row mask (test rows weigh 0) with its own 1/n_train scaling -- a *lane*;
  * every (fold, other-params) pair is one warm-started alpha path solved on the device;
  * up to sixteen lanes share each pass over X (``slm_solve_lanes``; sixteen in working-set solves on large
    X, otherwise as many as the fused kernel table has for this p); units are dealt fold-major so the
    lanes of a batch share one row mask and one working-set Gram;
  * for the ``Adaptive*`` estimators every (candidate, fold) re-weighting loop is a lane instead: outer
    iteration k of up to sixteen loops is one call, each lane with its own weight vectors, row mask and warm
    start (``_adaptive_lanes``; loop semantics of reference _adaptive_lasso.py:206-232);
  * ``fit_intercept=True`` is an unpenalised column of ones appended to the device copy (jointly
    minimising over it is per-fold centring);
  * hold-out scores come from the device as well (``slm_eval_sse_sparse`` on the gathered support
    columns of a unit's solutions, ``slm_eval_sse`` otherwise, with the test mask);
  * across processes (``torch.distributed`` launched one rank per GPU) the (fold, params) units are
    dealt to ranks with ``distributed.shard_units`` and gathered -- no data-path collective.

Anything else (``standardize=True`` and the overlap classes, which assemble another design matrix per
fit; sample weights, custom scorers, grids without ``alpha``) runs through scikit-learn's generic loop, still with the one-std rule.
"""

from __future__ import annotations

import numbers
import time
import warnings
from collections import defaultdict
from copy import deepcopy

import numpy as np
from sklearn.base import clone, is_classifier
from sklearn.model_selection import GridSearchCV as _GridSearchCV
from sklearn.model_selection import ParameterGrid, check_cv
from sklearn.model_selection._search import BaseSearchCV
from sklearn.utils.validation import check_is_fitted, indexable

from . import _backend, _engine
from . import distributed as D
from .model._adaptive_lasso import AdaptiveLasso
from .model._base import ProxRegressor

__all__ = ["GridSearchCV", "LineSearchCV"]

_FAST_SCORINGS = ("neg_root_mean_squared_error", "neg_mean_squared_error", "r2")


def select_best_index_onestd(results, refit_metric="score"):
    """One-standard-error rule (behaviour of reference model_selection.py:190-223).

    Start from the best-ranked candidate, mean score ``m*`` with standard deviation ``s*`` over the folds.
    Its "size" is the sum of its numerical hyper-parameters -- only the columns whose values are all numbers
    and all >= 0 (up to 1e-9) count: regularisation strengths.  Among the candidates at least that large the
    winner is the one whose mean score lies closest to ``m* - s*``: the most regularised model still within
    one standard error of the best."""
    mean = np.asarray(results[f"mean_test_{refit_metric}"], dtype=float)
    best = int(np.argmin(results[f"rank_test_{refit_metric}"]))
    target = mean[best] - results[f"std_test_{refit_metric}"][best]
    size = np.zeros(len(mean))
    for key, column in results.items():
        if not key.startswith("param_"):
            continue
        values = list(column)
        if not all(isinstance(v, numbers.Number) for v in values):
            continue
        values = np.asarray(values, dtype=float)
        if np.all(values > -1e-9):
            size += values
    eligible = np.flatnonzero(size >= size[best])
    return int(eligible[np.argmin(np.abs(mean[eligible] - target))])


class GridSearchCV(_GridSearchCV):
    """Exhaustive search over a parameter grid with optional one-standard-error selection.

    Args:
        estimator: estimator object (any scikit-learn estimator; the sparselm_amd Lasso family
            gets the device-resident fast path described in the module docstring).
        param_grid (dict | list[dict]): as in scikit-learn.
        opt_selection_method (str): "max_score" (default) or "one_std_score".
        scoring: default "neg_root_mean_squared_error" (reference :166).
        n_jobs, refit, cv, verbose, pre_dispatch, error_score, return_train_score: as scikit-learn.
        lanes (int): (fold, grid-row) units solved per pass over X on the fast path (1..16, default
            16; the engine falls back to fewer where no kernel variant serves that many).
        streams (int): engines (HIP streams) of the device the batches of the fast path are dealt to (default 1).
            Every further stream works on a device-to-device copy of the dataset, from its own host thread: the
            launches between the passes of one batch run beside the passes over X of another (the counterpart of
            ``n_jobs`` on one GPU).  Setting the copies up costs about what one batch of a 100 000 x 5 000
            problem does (allocations, the column-major copy): worth it for searches of many batches
            (tools/grid_streams_probe.py: 2 500 fits in 0.174 / 0.155 / 0.145 s on 1 / 2 / 3 standing streams).
        error_score: as scikit-learn; on the fast path a batch whose solve fails (non-finite iterate) scores
            ``error_score`` in its cells, or re-raises with ``error_score="raise"``.
    """

    def __init__(
        self,
        estimator,
        param_grid,
        *,
        opt_selection_method="max_score",
        scoring="neg_root_mean_squared_error",
        n_jobs=None,
        refit=True,
        cv=None,
        verbose=0,
        pre_dispatch="2*n_jobs",
        error_score=np.nan,
        return_train_score=False,
        lanes=16,
        streams=1,
    ):
        super().__init__(
            estimator=estimator,
            param_grid=param_grid,
            scoring=scoring,
            n_jobs=n_jobs,
            refit=refit,
            cv=cv,
            verbose=verbose,
            pre_dispatch=pre_dispatch,
            error_score=error_score,
            return_train_score=return_train_score,
        )
        self.opt_selection_method = opt_selection_method
        self.lanes = lanes
        self.streams = streams

    # ------------------------------------------------------------------------------------------
    def _fast_path_ok(self, fit_params) -> bool:
        est = self.estimator
        if not isinstance(est, ProxRegressor):
            return False
        if getattr(_backend.get_backend(), "name", None) != "hip":  # (tests may inject another backend)
            return False
        # standardize=True and the overlap classes assemble another design matrix on the host per fit
        if est._needs_host_preprocessing() or fit_params:
            return False
        if self.scoring not in _FAST_SCORINGS or self.return_train_score:
            return False
        if self.refit not in (True, False):
            return False
        grids = self.param_grid if isinstance(self.param_grid, (list, tuple)) else [self.param_grid]
        return all(isinstance(g, dict) and "alpha" in g for g in grids)

    def fit(self, X, y=None, *, groups=None, **fit_params):
        """Run the search (reference model_selection.py:226-424)."""
        if self.opt_selection_method not in ("max_score", "one_std_score"):
            raise ValueError(f"opt_selection_method {self.opt_selection_method!r} is not supported")
        if self._fast_path_ok(fit_params):
            return self._fit_device(X, y, groups)
        return self._fit_generic(X, y, groups, fit_params)

    # ---- generic: scikit-learn's loop + selection rule ------------------------------------------
    def _fit_generic(self, X, y, groups, fit_params):
        user_refit = self.refit
        onestd = self.opt_selection_method == "one_std_score"
        if onestd and user_refit is True:
            self.refit = lambda results: select_best_index_onestd(results)
        try:
            super().fit(X, y, groups=groups, **fit_params)
        finally:
            self.refit = user_refit
        # the rule applies to every single-metric search, refitted or not (reference :356-372): with
        # refit=False (LineSearchCV lines do that) best_params_ must still be the one-std choice
        if onestd and not self.multimetric_ and not (user_refit is True):
            if isinstance(user_refit, str) or not user_refit:
                self.best_index_ = select_best_index_onestd(self.cv_results_)
                self.best_params_ = self.cv_results_["params"][self.best_index_]
        if hasattr(self, "best_index_"):
            self.best_score_ = self.cv_results_["mean_test_score"][self.best_index_]
            self.best_score_std_ = self.cv_results_["std_test_score"][self.best_index_]
        return self

    # ---- device-resident fast path ------------------------------------------------------------------
    def _fit_device(self, X, y, groups):
        est = self.estimator
        X, y, groups = indexable(X, y, groups)
        X = np.asarray(X, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        n, p = X.shape
        cv = check_cv(self.cv, y, classifier=is_classifier(est))
        splits = list(cv.split(X, y, groups))
        n_splits = len(splits)
        candidates = list(ParameterGrid(self.param_grid))
        # validate the way fit() would (same error classes): every distinct value of every grid parameter once
        # (the constraints are per parameter; 60 clones instead of 500 on a 50 x 10 grid, whose
        # clone / get_params / inspect.signature cost was a quarter of the search)
        seen = set()
        for params in candidates:
            fresh = [k for k, v in params.items() if (k, repr(v)) not in seen]
            if fresh:
                seen.update((k, repr(v)) for k, v in params.items())
                clone(est).set_params(**params)._validate_params(X, y)

        # units: (non-alpha params, fold) -> one warm-started alpha path
        by_combo = defaultdict(list)
        for ci, params in enumerate(candidates):
            key = tuple(sorted((k, repr(v)) for k, v in params.items() if k != "alpha"))
            by_combo[key].append(ci)
        adaptive = isinstance(est, AdaptiveLasso)
        if adaptive:  # every (candidate, fold) is its own re-weighting loop: no shared alpha path
            by_combo = {ci: [ci] for ci in range(len(candidates))}
        combos = list(by_combo.values())
        # fold-major: the units of one batch then mostly share a fold, i.e. one row mask, and the
        # engine builds ONE working-set Gram for all lanes with the same mask (same host array)
        units = [(c, f) for f in range(n_splits) for c in range(len(combos))]
        train_masks, test_masks = [], []
        for train, test in splits:
            m = np.zeros(n)
            m[train] = 1.0
            train_masks.append(m)
            t = np.zeros(n)
            t[test] = 1.0
            test_masks.append(t)

        rank, world = D.active_world()
        eng = _engine.get_engine()
        t0 = time.perf_counter()
        scores = np.full((len(candidates), n_splits), np.nan)
        fit_time = np.zeros((len(candidates), n_splits))
        # fit_intercept=True: the intercept is an unpenalised coefficient on a column of ones appended to
        # the device copy (its own group, zero weights).  Minimising over it jointly is what centring X and
        # y by their training-fold means does (reference _base.py:207-227), fold by fold, without a
        # centred copy per fold.
        intercept = bool(est.fit_intercept)
        Xd = np.hstack([X, np.ones((n, 1))]) if intercept else X

        def with_intercept(a, b, d, G):
            """penalty vectors of the augmented problem (None stays None: that term is off)"""
            if not intercept:
                return a, b, d
            a = None if a is None else np.append(np.broadcast_to(a, (p,)), 0.0)
            b = None if b is None else np.append(np.broadcast_to(b, (G,)), 0.0)
            d = None if d is None else np.append(np.broadcast_to(d, (G,)), 0.0)
            return a, b, d

        with eng.dataset(Xd, y) as ds:
            base = clone(est).set_params(alpha=1.0)
            a1, b1, d1, gidx, G = base._penalty(X)
            if gidx is not None:
                ds.set_groups(np.append(gidx, G) if intercept else gidx, G + 1 if intercept else G)
            my_units = [units[i] for i in D.shard_units(len(units), rank, world)]
            lanes = max(1, min(int(self.lanes), _engine.MAX_LANES, ds.max_lanes()))
            opts = _solver_options(est)
            opts.setdefault("tol", _backend.default_tol(n, p))
            local = {}
            batches = [my_units[k0 : k0 + lanes] for k0 in range(0, len(my_units), lanes)]

            def run_batch(ds, batch):
                """One call of the engine for the units of `batch` on dataset `ds` (this stream's copy); the scores
                go into `local`, the return value counts solves that stopped short of the tolerance."""
                if adaptive:
                    ests = [clone(est).set_params(**candidates[combos[c][0]]) for c, _ in batch]
                    t_batch = time.perf_counter()
                    try:
                        fits = _adaptive_lanes(ds, ests, X, [train_masks[f] for _, f in batch],
                                               [len(splits[f][0]) for _, f in batch], opts, with_intercept)
                    except _engine.NonFiniteError:
                        if self.error_score == "raise":
                            raise
                        for c, f in batch:
                            local[(c, f)] = (combos[c], np.full(len(combos[c]), self.error_score, dtype=float), 0.0)
                        return 0
                    dt = (time.perf_counter() - t_batch) / len(batch)
                    for (c, f), fit in zip(batch, fits):
                        sse = ds.eval_sse(fit["beta"][None, :], test_masks[f])
                        local[(c, f)] = (combos[c], self._score_from_sse(sse, y[splits[f][1]]), dt)
                    return sum(not i["converged"] for fit in fits for i in fit["infos"])
                # a batch with spare lane slots (the last one; every one when there are fewer units than
                # lanes, e.g. a grid dealt over 8 GPUs) cuts each unit's path into contiguous ranges, one
                # lane each: a pass advances every lane by one point, so the call needs K / split passes
                specs, metas = [], []
                for c, f in batch:
                    cis = sorted(combos[c], key=lambda ci: -candidates[ci]["alpha"])
                    e = clone(est).set_params(**{k: v for k, v in candidates[cis[0]].items() if k != "alpha"})
                    e.set_params(alpha=1.0)
                    a, b, d, _, G_e = e._penalty(X)
                    a, b, d = with_intercept(a, b, d, G_e if G_e is not None else p)
                    alphas = np.array([candidates[ci]["alpha"] for ci in cis], dtype=float)
                    pts = np.c_[
                        alphas if a is not None else 0 * alphas,
                        alphas if b is not None else 0 * alphas,
                        np.ones_like(alphas) if d is not None else 0 * alphas,
                    ]
                    train, test = splits[f]
                    split = max(1, min(lanes // len(batch), len(cis) // 4)) if lanes >= 8 else 1
                    for part in np.array_split(np.arange(len(cis)), split):
                        specs.append(dict(points=pts[part], a=a, b=b, d=d, row_weight=train_masks[f], n_eff=len(train)))
                    metas.append((cis, test, split))
                t_batch = time.perf_counter()
                try:
                    results = _solve_lanes_with_fallback(ds, specs, opts)
                except _engine.NonFiniteError:  # the counterpart of a failing fit in _fit_and_score
                    if self.error_score == "raise":
                        raise
                    for (c, f), (cis, _, _) in zip(batch, metas):
                        local[(c, f)] = (cis, np.full(len(cis), self.error_score, dtype=float), 0.0)
                    return 0
                dt = (time.perf_counter() - t_batch) / max(1, sum(len(m[0]) for m in metas))
                at = 0
                for (c, f), (cis, test, split) in zip(batch, metas):
                    betas = np.vstack([r.betas for r in results[at : at + split]])
                    at += split
                    sse = ds.eval_sse(betas, test_masks[f])
                    local[(c, f)] = (cis, self._score_from_sse(sse, y[test]), dt)
                return sum(not r.converged for r in results)

            unconverged = self._run_batches(ds, batches, run_batch, gidx, G, intercept)
            merged = _gather(local, units, world)
            for (c, f), (cis, sc, dt) in merged.items():
                scores[cis, f] = sc
                fit_time[cis, f] = dt
            if unconverged:
                self._warn_unconverged(unconverged, "grid cell solves", opts)

            self.cv_results_ = _format_results(candidates, scores, fit_time)
            self.n_splits_ = n_splits
            self.multimetric_ = False
            self.scorer_ = self.scoring
            if self.opt_selection_method == "one_std_score":
                self.best_index_ = select_best_index_onestd(self.cv_results_)
            else:
                self.best_index_ = int(self.cv_results_["rank_test_score"].argmin())
            self.best_params_ = candidates[self.best_index_]
            self.best_score_ = self.cv_results_["mean_test_score"][self.best_index_]
            self.best_score_std_ = self.cv_results_["std_test_score"][self.best_index_]
            if self.refit:
                t1 = time.perf_counter()
                best = clone(est).set_params(**self.best_params_)
                if adaptive:
                    fit = _adaptive_lanes(ds, [best], X, [None], [n], opts, with_intercept)[0]
                    beta_aug = fit["beta"]
                    best.n_iter_ = fit["n_iter"]
                    best.adaptive_weights_ = fit["weights"]
                    best.solver_info_ = {"solves": fit["infos"]}
                    if not all(i["converged"] for i in fit["infos"]):
                        self._warn_unconverged(1, "the refit", opts)
                else:
                    a, b, d, _, G_b = best._penalty(X)
                    a, b, d = with_intercept(a, b, d, G_b if G_b is not None else p)
                    res = ds.solve_path(
                        [(1.0, 1.0, 1.0)],
                        a=np.zeros(ds.p) if a is None else a,
                        b=np.zeros(ds.n_groups) if b is None else b,
                        d=np.zeros(ds.n_groups) if d is None else d,
                        **opts,
                    )
                    beta_aug = res.betas[0].copy()
                    best.solver_info_ = {"n_iter": int(res.n_iter[0]), "converged": res.converged}
                    if not res.converged:
                        self._warn_unconverged(1, "the refit", opts)
                best.coef_ = beta_aug[:p].copy()
                best.intercept_ = float(beta_aug[p]) if intercept else 0.0
                best.n_features_in_ = p
                self.best_estimator_ = best
                self.refit_time_ = time.perf_counter() - t1
        self.search_time_ = time.perf_counter() - t0
        return self

    def _run_batches(self, ds, batches, run_batch, gidx, G, intercept):
        """Deal the batches to ``streams`` engines of the device: the first works on `ds` itself from this thread,
        every further one on a device-to-device copy from a thread of its own (the engine calls release the GIL).
        One stream, or fewer batches than two: plain loop."""
        streams = max(1, min(int(self.streams), len(batches)))
        if streams == 1:
            return sum(run_batch(ds, batch) for batch in batches)
        import threading

        copies = [ds]
        try:
            for _ in range(streams - 1):
                c = ds.clone()
                copies.append(c)
                if gidx is not None:
                    c.set_groups(np.append(gidx, G) if intercept else gidx, G + 1 if intercept else G)
            todo = list(range(len(batches)))
            lock = threading.Lock()
            counts, errors = [0] * streams, []

            def work(i):
                try:
                    while True:
                        with lock:
                            if not todo or errors:
                                return
                            b = todo.pop(0)
                        counts[i] += run_batch(copies[i], batches[b])
                except BaseException as exc:  # re-raised by the caller's thread
                    with lock:
                        errors.append(exc)

            threads = [threading.Thread(target=work, args=(i,)) for i in range(1, streams)]
            for t in threads:
                t.start()
            work(0)
            for t in threads:
                t.join()
            if errors:
                raise errors[0]
            return sum(counts)
        finally:
            for c in copies[1:]:
                eng = c.engine
                c.close()
                eng.close()

    @staticmethod
    def _warn_unconverged(count, what, opts):
        from sklearn.exceptions import ConvergenceWarning

        warnings.warn(
            f"{count} of {what} stopped at max_iter={opts.get('max_iter', 10000)} before reaching "
            f"tol={opts.get('tol')}: the scores / coefficients come from unconverged iterates; increase "
            "solver_options['max_iter'].",
            ConvergenceWarning,
        )

    def _score_from_sse(self, sse, y_test):
        mse = sse / len(y_test)
        if self.scoring == "neg_mean_squared_error":
            return -mse
        if self.scoring == "neg_root_mean_squared_error":
            return -np.sqrt(mse)
        sst = float(np.sum((y_test - np.mean(y_test)) ** 2))
        return 1.0 - sse / sst if sst > 0 else np.where(sse == 0, 1.0, 0.0)

    def predict(self, X):
        check_is_fitted(self, "best_estimator_")
        return self.best_estimator_.predict(X)


class LineSearchCV(BaseSearchCV):
    """Cyclic one-dimensional grid searches (reference model_selection.py:427-707).

    ``param_grid`` is a list of ``(name, values)`` pairs; iteration i searches parameter
    ``i % n_params`` over its values with every other parameter fixed at its current best (initially
    the first value of its list); ``n_iter`` defaults to ``2 * n_params`` (:650-654).  Every line is a
    ``GridSearchCV`` of this module, so alpha lines of the Lasso family run on the device-resident
    fast path.  After ``fit`` the fitted attributes of the last line search are exposed (:695-703) and
    ``history_`` holds all of them.
    """

    def __init__(
        self,
        estimator,
        param_grid,
        *,
        opt_selection_method="max_score",
        n_iter=None,
        scoring="neg_root_mean_squared_error",
        n_jobs=None,
        refit=True,
        cv=None,
        verbose=0,
        pre_dispatch="2*n_jobs",
        error_score=np.nan,
        return_train_score=False,
    ):
        super().__init__(
            estimator=estimator,
            scoring=scoring,
            n_jobs=n_jobs,
            refit=refit,
            cv=cv,
            verbose=verbose,
            pre_dispatch=pre_dispatch,
            error_score=error_score,
            return_train_score=return_train_score,
        )
        self.param_grid = param_grid
        self.opt_selection_method = opt_selection_method
        self.n_iter = n_iter

    def fit(self, X, y=None, *, groups=None, **fit_params):
        if not (
            isinstance(self.param_grid, (list, tuple))
            and len(self.param_grid) > 0
            and isinstance(self.param_grid[0], (tuple, list))
            and isinstance(self.param_grid[0][0], str)
        ):
            raise ValueError("Parameter grid is not given in the correct format!")
        n_params = len(self.param_grid)
        if self.opt_selection_method is None:
            methods = ["max_score"] * n_params
        elif isinstance(self.opt_selection_method, str):
            methods = [self.opt_selection_method] * n_params
        elif (
            isinstance(self.opt_selection_method, (list, tuple))
            and all(isinstance(m, str) for m in self.opt_selection_method)
            and len(self.opt_selection_method) == n_params
        ):
            methods = list(self.opt_selection_method)
        else:
            raise ValueError(
                "Optimal hyperparams selection methods should be given as a"
                " single string, or as a list of strings with the same"
                " amount of parameters!"
            )
        n_iter = self.n_iter if (self.n_iter is not None and self.n_iter > 0) else 2 * n_params
        history = []
        best = None
        for i in range(n_iter):
            pid = i % n_params
            last = [values[0] if best is None else best[name] for name, values in self.param_grid]
            line = {
                name: (list(values) if k == pid else [lv])
                for k, ((name, values), lv) in enumerate(zip(self.param_grid, last))
            }
            search = GridSearchCV(
                estimator=self.estimator,
                param_grid=line,
                opt_selection_method=methods[pid],
                scoring=self.scoring,
                n_jobs=self.n_jobs,
                refit=self.refit,
                cv=self.cv,
                verbose=self.verbose,
                pre_dispatch=self.pre_dispatch,
                error_score=self.error_score,
                return_train_score=self.return_train_score,
            )
            search.fit(X, y, groups=groups, **fit_params)
            best = deepcopy(search.best_params_)
            history.append(search)
        self.history_ = history
        for attr in (v for v in vars(history[-1]) if v.endswith("_") and not v.startswith("__")):
            setattr(self, attr, getattr(history[-1], attr))
        return self

    def _run_search(self, evaluate_candidates):
        """Unused: every line is its own GridSearchCV."""
        return


def _solver_options(est) -> dict:
    """``solver_options`` as keyword arguments of ``Dataset.solve_lanes`` / ``solve_path`` -- the same
    options, with the same meaning, as ``_backend.SolveProblem.solve`` gives a plain ``fit``."""
    from ._backend import normalise_options

    o = normalise_options(est.solver_options)
    out = {}
    if "tol" in o:
        out["tol"] = float(o["tol"])
    if "max_iter" in o:
        out["max_iter"] = int(o["max_iter"])
    if "L" in o:
        out["L"] = float(o["L"])
    if "check_every" in o:
        out["check_every"] = int(o["check_every"])
    if not o.get("restart", True):
        out["flags"] = _engine.FLAG_NO_RESTART
    return out


def _solve_lanes_with_fallback(ds, specs, opts):
    """As many lanes as asked for when a kernel variant serves them for this p, otherwise halves."""
    try:
        return ds.solve_lanes(specs, **opts)
    except NotImplementedError:
        if len(specs) == 1:
            raise
        half = len(specs) // 2
        return _solve_lanes_with_fallback(ds, specs[:half], opts) + _solve_lanes_with_fallback(ds, specs[half:], opts)


def _adaptive_lanes(ds, ests, X, row_weights, n_effs, opts, with_intercept):
    """The re-weighting loops of several Adaptive* estimators side by side: outer iteration k of every
    estimator is ONE call with one lane per estimator (each lane its own weight vectors, row mask and warm
    start), so X is read once per inner iteration for all of them.  Same loop semantics as
    AdaptiveLasso._solve (reference _adaptive_lasso.py:206-232): weights updated after every solve, early
    stop per estimator on ``||w_new - w_prev|| <= tol``, coefficients of the last solve returned.

    Returns one dict per estimator: ``beta`` (augmented with the intercept coefficient when the dataset has
    the column of ones), ``n_iter``, ``weights``, ``infos``.
    """
    p = X.shape[1]
    st = []
    for e in ests:
        if e.max_iter < 1:
            raise ValueError("max_iter=0 performs no solve; coef_ would be undefined")
        _, G, w = e._adaptive_setup(X)
        st.append(dict(G=G, w=w, prev=w.copy(), beta=None, n_iter=0, done=False, infos=[]))
    for it in range(max(e.max_iter for e in ests)):
        live = [i for i, e in enumerate(ests) if not st[i]["done"] and it < e.max_iter]
        if not live:
            break
        specs = []
        for i in live:
            a, b, d = with_intercept(*ests[i]._weights_to_penalty(st[i]["w"], p, st[i]["G"]), st[i]["G"])
            specs.append(dict(points=[(1.0, 1.0, 1.0)], a=a, b=b, d=d,
                              beta0=st[i]["beta"] if ests[i].warm_start else None,
                              row_weight=row_weights[i], n_eff=n_effs[i]))
        results = _solve_lanes_with_fallback(ds, specs, dict(opts, want_group_norms=True))
        for i, res in zip(live, results):
            s = st[i]
            s["beta"] = res.betas[0].copy()
            s["n_iter"] += 1
            s["infos"].append({"n_iter": int(res.n_iter[0]), "converged": res.converged})
            w = ests[i]._updated_weights(s["beta"][:p], res.group_norms[0][: s["G"]])
            s["done"] = bool(np.linalg.norm(w - s["prev"]) <= ests[i].tol)
            s["prev"], s["w"] = w.copy(), w
    return [dict(beta=s["beta"], n_iter=s["n_iter"], weights=s["w"], infos=s["infos"]) for s in st]


def _gather(local: dict, units, world: int = 1) -> dict:
    """Every rank's cells on every rank; the check that no unit is missing runs whatever the world size."""
    merged = dict(local)
    if world > 1:
        import torch.distributed as dist  # (only a multi-rank search needs torch at all)

        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, local)
        merged = {}
        for part in parts:
            merged.update(part)
    missing = [u for u in units if u not in merged]
    if missing:
        raise RuntimeError(f"grid units {missing[:4]} were not solved by any rank")
    return merged


def _format_results(candidates, scores, fit_time) -> dict:
    """cv_results_ in scikit-learn's layout (BaseSearchCV._format_results)."""
    from scipy.stats import rankdata

    n_cand, n_splits = scores.shape
    results = {}
    results["mean_fit_time"] = fit_time.mean(axis=1)
    results["std_fit_time"] = fit_time.std(axis=1)
    results["mean_score_time"] = np.zeros(n_cand)
    results["std_score_time"] = np.zeros(n_cand)
    names = sorted({k for c in candidates for k in c})
    for name in names:
        vals = np.ma.MaskedArray(np.empty(n_cand, dtype=object), mask=True)
        for i, c in enumerate(candidates):
            if name in c:
                vals[i] = c[name]
        try:
            if not vals.mask.any():
                vals = np.ma.MaskedArray(np.array([c[name] for c in candidates]), mask=False)
        except (ValueError, TypeError):
            pass
        results[f"param_{name}"] = vals
    results["params"] = candidates
    for f in range(n_splits):
        results[f"split{f}_test_score"] = scores[:, f]
    mean = scores.mean(axis=1)
    results["mean_test_score"] = mean
    results["std_test_score"] = scores.std(axis=1)
    if np.isnan(mean).all():
        results["rank_test_score"] = np.ones(n_cand, dtype=np.int32)
    else:
        min_score = np.nanmin(mean) - 1
        results["rank_test_score"] = rankdata(-np.nan_to_num(mean, nan=min_score), method="min").astype(np.int32)
    return results
