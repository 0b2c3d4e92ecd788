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
import os
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
# What a grid of the device path may vary: parameters that only enter through `_penalty` (evaluated per unit on a clone
# with the candidate's values) and, for the Adaptive* classes, the controls of the re-weighting loop (every candidate runs
# its own).  Anything else -- `standardize`, `fit_intercept`, `groups`, `solver_options`, ... -- changes the design, the
# group structure on the device or the preprocessing, which the device path sets up ONCE from the base estimator: such
# grids go through scikit-learn's generic loop.
_PENALTY_PARAMS = frozenset({"alpha", "l1_ratio", "delta", "group_weights"})
_ADAPTIVE_LOOP_PARAMS = frozenset({"max_iter", "eps", "tol", "update_function", "warm_start"})


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


class _DatasetLease:
    """The device dataset of a search, kept across the searches of a `LineSearchCV`: every line is a `GridSearchCV` on
    the same (X, y), and opening a dataset per line uploaded X once per line (4 GB at the BASELINE shape: as long as the
    line's search itself).  The lease opens it at the first line and closes it when the line search ends; what was built on
    it in between -- the column copy of the split pass, the Grams of covariance passes -- stays."""

    def __init__(self, repeats=1):
        self.ds, self.key, self.repeats = None, None, int(repeats)

    def get(self, grid):
        key = (id(grid.X_in), id(grid.y_in), grid.intercept, grid.X.shape)
        if self.ds is None or self.key != key:
            self.close()
            self.ds, self.key = grid.open(), key
        else:
            grid.adopt(self.ds)
        return self.ds

    def close(self):
        if self.ds is not None:
            self.ds.close()
            self.ds, self.key = None, None


class _Borrowed:
    """`with` wrapper that leaves the dataset open (the lease closes it)."""

    def __init__(self, ds):
        self.ds = ds

    def __enter__(self):
        return self.ds

    def __exit__(self, *exc):
        return False


class _Cached:
    """`with` wrapper of a dataset taken from the device dataset cache (`_backend.DatasetCache`: keyed by the content of
    what was uploaded): handed back, not closed -- a second search on the same small (X, y), the README's workflow of
    trying grids on one dataset, finds it on the device (opening and destroying a dataset is a few dozen allocations: 1.1 ms
    of a 10 ms search at the reference's sizes)."""

    def __init__(self, item):
        self.item = item

    def __enter__(self):
        return self.item[0]

    def __exit__(self, *exc):
        _backend.dataset_cache().release(*self.item)
        return False


class GridSearchCV(_GridSearchCV):
    """Exhaustive search over a parameter grid with optional one-standard-error selection.

    Args:
        estimator: estimator object (any scikit-learn estimator; the sparselm_amd Lasso family
            gets the device-resident fast path described in the module docstring).
        param_grid (dict | list[dict]): as in scikit-learn.
        opt_selection_method (str): "max_score" (default) or "one_std_score".
        scoring: default "neg_root_mean_squared_error" (reference :166).
        n_jobs, refit, cv, verbose, pre_dispatch, error_score, return_train_score: as scikit-learn.
        lanes (int | None): (fold, grid-row) units solved per call on the fast path.  None (default): as many as the dataset
            takes -- sixteen per pass over a large X, up to 64 cells per launch where the on-chip solver applies (the
            reference's README search: its 50 (candidate, fold) cells in ONE call per re-weighting round); an int caps it.
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
        lanes=None,
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
        allowed = _PENALTY_PARAMS | (_ADAPTIVE_LOOP_PARAMS if isinstance(est, AdaptiveLasso) else frozenset())
        return all(isinstance(g, dict) and "alpha" in g and set(g) <= allowed for g in grids)

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
        rank, world = D.active_world()
        # among several ranks the search runs on an engine of its own that carries a communicator for the folds' Grams
        # (None where RCCL cannot form the group: the search then goes without sharded Grams)
        grid = _DeviceGrid(self, X, y, groups, engine=D.grid_engine(rank, world) if world > 1 else None)
        t0 = time.perf_counter()
        lease = getattr(self, "_lease", None)
        with (_Borrowed(lease.get(grid)) if lease is not None else grid.open_cached()) as ds:
            local, unconverged = grid.solve_share(ds, rank, world)
            scores, fit_time = grid.merge(_gather(local, grid.cells, world))
            if unconverged:
                self._warn_unconverged(unconverged, "grid cell solves", grid.opts)
            candidates = grid.candidates
            self.cv_results_ = _format_results(candidates, scores, fit_time)
            self.n_splits_ = grid.n_splits
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
                self.best_estimator_ = grid.refit(ds, self.best_params_)
                self.refit_time_ = time.perf_counter() - t1
        self.search_time_ = time.perf_counter() - t0
        return self

    def _device_cells(self, X, y, groups=None, rank=0, world=1, engine=None):
        """The (candidate, fold) scores rank ``rank`` of a ``world``-rank search computes, as a (candidates x folds)
        array with NaN in the cells of the other ranks -- the share `_fit_device` would solve in that process, without a
        process group (tests; `bench.py`'s emulated multi-rank legs).  ``engine``: the rank's engine when the ranks are
        the engines of an in-process communicator (``_engine.init_local_comm``; one thread per rank must then call this
        at the same time: the folds' Grams are a collective)."""
        grid = _DeviceGrid(self, X, y, groups, engine=engine)
        with grid.open() as ds:
            local, _ = grid.solve_share(ds, rank, world)
        scores = np.full((len(grid.candidates), grid.n_splits), np.nan)
        for cis, f, sc, _ in local.values():
            scores[cis, f] = sc
        return scores

    def _run_batches(self, ds, batches, run_batch, grid):
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
                if grid.gidx is not None:
                    grid._set_groups(c)
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


from ._backend import normalise_options  # noqa: E402


class _DeviceGrid:
    """One device-resident search: the grid laid out as units, the dataset, the calls of a rank's share.

    A *unit* is one warm-started alpha path: a (fold, other-parameters) pair -- or, for the Adaptive* estimators, one
    (candidate, fold) re-weighting loop.  The POINTS of the units' paths are dealt to the lane slots of all ranks by
    `distributed.plan_lane_calls`; what every rank reports back are *cells*: (unit, first point) -> (candidate
    indices, fold, scores, seconds per fit)."""

    def __init__(self, search, X, y, groups, engine=None):
        est = search.estimator
        self.search, self.est = search, est
        self.engine = engine  # None: the per-process default engine
        self.X_in, self.y_in = X, y  # (the caller's objects: what a lease recognises the next line's data by)
        X, y, groups = indexable(X, y, groups)
        self.X = X = np.asarray(X, dtype=np.float64)
        self.y = y = np.asarray(y, dtype=np.float64)
        if not np.all(np.isfinite(y)):  # (X is scanned on the device copy, open())
            raise ValueError("Input y contains NaN or infinity.")
        n, p = X.shape
        cv = check_cv(search.cv, y, classifier=is_classifier(est))
        self.splits = splits = list(cv.split(X, y, groups))
        self.n_splits = len(splits)
        self.candidates = candidates = list(ParameterGrid(search.param_grid))
        # validate the way fit() would (same error classes): every distinct value of every grid parameter once
        # (the constraints are per parameter; 60 clones instead of 500 on a 50 x 10 grid, whose
        # clone / get_params / inspect.signature cost was a quarter of the search)
        seen = set()
        for params in candidates:
            fresh = [k for k, v in params.items() if (k, _value_key(v)) not in seen]
            if fresh:
                seen.update((k, _value_key(v)) for k, v in params.items())
                clone(est).set_params(**params)._validate_params(X, y)
        # units: (non-alpha params, fold) -> one warm-started alpha path
        by_combo = defaultdict(list)
        for ci, params in enumerate(candidates):
            key = tuple(sorted((k, _value_key(v)) for k, v in params.items() if k != "alpha"))
            by_combo[key].append(ci)
        self.adaptive = isinstance(est, AdaptiveLasso)
        if self.adaptive:  # every (candidate, fold) is its own re-weighting loop: no shared alpha path
            by_combo = {ci: [ci] for ci in range(len(candidates))}
        # (the candidates of a combination in path order: alpha descending)
        self.combos = [sorted(cis, key=lambda ci: -candidates[ci]["alpha"]) for cis in by_combo.values()]
        # fold-major: the units of one call then mostly share a fold, i.e. one row mask, and the
        # engine builds ONE working-set Gram for all lanes with the same mask (same host array)
        self.units = [(c, f) for f in range(self.n_splits) for c in range(len(self.combos))]
        self.train_masks, self.test_masks = [], []
        for train, test in splits:
            m = np.zeros(n)
            m[train] = 1.0
            self.train_masks.append(m)
            t = np.zeros(n)
            t[test] = 1.0
            self.test_masks.append(t)
        # fit_intercept=True: the intercept is an unpenalised coefficient on a column of ones appended to
        # the device copy (its own group, zero weights).  Minimising over it jointly is what centring X and
        # y by their training-fold means does (reference _base.py:207-227), fold by fold, without a
        # centred copy per fold.
        self.intercept = bool(est.fit_intercept)
        base = clone(est).set_params(alpha=1.0)
        _, _, _, self.gidx, self.G = base._penalty(X)
        self.opts = _solver_options(est)
        self.opts.setdefault("tol", _backend.default_tol(n, p))
        self._pen = {}
        # every (unit, first point index of a piece) some rank has to report (`_gather` checks that none is missing)
        self.cells = None

    def open(self):
        n = self.X.shape[0]
        Xd = np.hstack([self.X, np.ones((n, 1))]) if self.intercept else self.X
        eng = self.engine if self.engine is not None else _engine.get_engine()
        ds = eng.dataset(Xd, self.y)
        _backend.raise_if_nonfinite(ds)  # (what every cell's fit would raise: scikit-learn's ValueError, from a scan of the device copy)
        if eng.comm_ranks() > 1:  # grid mode among ranks: every rank holds all rows
            ds.set_replicated(True)
        if self.gidx is not None:
            self._set_groups(ds)
        self.lanes = self._lanes_for(ds)
        self._plan_world = None
        return ds

    def open_cached(self):
        """`open()` through the device dataset cache where it applies (the default engine, matrices the cache takes);
        a context manager either way."""
        if self.engine is not None:
            return self.open()
        n = self.X.shape[0]
        Xd = np.hstack([self.X, np.ones((n, 1))]) if self.intercept else self.X
        item = _backend.dataset_cache().acquire(_engine.get_engine(), Xd, self.y, None, False, check_finite=True)
        if item[3] is None:  # (too large for the cache: the search owns the dataset)
            return self.adopt(item[0])
        try:
            self.adopt(item[0])
        except BaseException:  # (e.g. set_groups refused the labels: the entry must not stay marked as in use for ever)
            _backend.dataset_cache().release(*item)
            raise
        return _Cached(item)

    def adopt(self, ds):
        """`open()` for a dataset another search of the same data has opened (a `_DatasetLease`)."""
        if self.gidx is not None:
            self._set_groups(ds)
        else:
            ds.set_groups(None)
        self.lanes = self._lanes_for(ds)
        self._plan_world = None
        return ds

    def _lanes_for(self, ds):
        # (the cap the ranks PLAN with is the smallest any of them has: a rank without the memory for the column-major copy
        #  serves sixteen lanes where its peers serve thirty-two, and ranks that plan on different caps deal the units differently)
        cap = D.min_over_ranks(ds.max_lanes(self.opts.get("flags", 0)))
        return cap if self.search.lanes is None else max(1, min(int(self.search.lanes), cap))

    def _set_groups(self, ds):
        ds.set_groups(np.append(self.gidx, self.G) if self.intercept else self.gidx, self.G + 1 if self.intercept else self.G)

    def with_intercept(self, a, b, d, G):
        """penalty vectors of the augmented problem (None stays None: that term is off)"""
        if not self.intercept:
            return a, b, d
        p = self.X.shape[1]
        a = None if a is None else np.append(np.broadcast_to(a, (p,)), 0.0)
        b = None if b is None else np.append(np.broadcast_to(b, (G,)), 0.0)
        d = None if d is None else np.append(np.broadcast_to(d, (G,)), 0.0)
        return a, b, d

    def penalty_of(self, c):
        """Combination c's penalty at alpha = 1 as (unit-scaled vectors, scales, key): ``a = sa * a_unit`` etc. with
        max |a_unit| = 1, so that combinations whose vectors are proportional -- the rows of an l1_ratio grid -- bring
        the SAME vectors and differ in the scales their path points carry; pieces of such units can follow each other
        in one lane (`key` is equal exactly then)."""
        hit = self._pen.get(c)
        if hit is None:
            cand = self.candidates[self.combos[c][0]]
            e = clone(self.est).set_params(**{k: v for k, v in cand.items() if k != "alpha"})
            e.set_params(alpha=1.0)
            a, b, d, _, G_e = e._penalty(self.X)
            vecs, scales, key = [], [], []
            for v in self.with_intercept(a, b, d, G_e if G_e is not None else self.X.shape[1]):
                if v is None:
                    vecs.append(None); scales.append(0.0); key.append(b"-")
                    continue
                v = np.asarray(v, dtype=np.float64)
                top = float(np.max(np.abs(v))) if v.size else 0.0
                if top > 0.0 and np.isfinite(top):
                    vecs.append(v / top); scales.append(top); key.append(np.round(v / top, 12).tobytes())
                else:
                    vecs.append(v); scales.append(1.0); key.append(b"0")
            hit = self._pen[c] = (vecs, scales, tuple(key))
        return hit

    def plan(self, world):
        """plan[rank] = calls (lists of lanes, a lane a list of (unit, point indices)), for all ranks"""
        if self._plan_world != world:
            if self.adaptive:  # one point per unit: a call is `lanes` re-weighting loops side by side
                mine = [D.shard_units(len(self.units), r, world) for r in range(world)]
                self._plan = [[[[(u, [0])] for u in own[k0 : k0 + self.lanes]] for k0 in range(0, len(own), self.lanes)]
                              for own in mine]
            else:
                keys = [(f, self.penalty_of(c)[2]) for c, f in self.units]
                # a lane's cold start costs tens of passes without the working set (the fused kernels' few lanes):
                # paths are only cut into pieces where sixteen lanes share a pass
                self._plan = D.plan_lane_calls([len(self.combos[c]) for c, _ in self.units], keys, world, self.lanes,
                                               fine=self.lanes >= 8)
            self._plan_world = world
            self.cells = [(u, idx[0]) for calls in self._plan for call in calls for lane in call for u, idx in lane]
        return self._plan

    def covariance(self, ds, calls, world=1):
        """Covariance passes for this search (``solver_options={"covariance": True | False | "auto"}``, default "auto"):
        the Gram of every fold's training rows is built once (``Dataset.covariance_folds``: where the folds' test rows
        partition the rows -- K-fold -- the Gram of all rows is the sum of the test rows' Grams, one triangle product over n
        rows in all) and every pass of a share reads 8 p^2 bytes per fold instead of X.  Among ranks (a replica on an
        engine with a communicator) every rank builds the parts of its ``world``-th of the rows and the ranks sum them: the
        decision is taken from ``calls``, the LARGEST share of the plan, so that every rank takes the same one (the build is
        a collective).  "auto" asks whether what the passes of the largest share save -- one per path point and sixteen
        lanes; a read of X at the HBM rate, and in a grid cut into pieces the dearer appends to the working set, against a
        read of the Gram -- exceeds the Grams by a margin: true for BASELINE config 4 on one GPU (2 500 fits: 0.147 s over
        X, 0.065 s of Grams + 0.055 s from them) and, with the build shared, for its eighth on one of eight ranks; not for
        splits that are no partition unless the grid is several times larger."""
        want = normalise_options(self.est.solver_options).get("covariance", "auto")
        if want is False or len(self.train_masks) > _engine.MAX_LANES:  # (a dataset keeps sixteen Grams)
            return False
        n, p = ds.n, ds.p
        eng = getattr(ds, "engine", None)
        shared = world > 1 and eng is not None and eng.comm_ranks() == world  # the ranks build the Grams together
        n_effs = [int(m.sum()) for m in self.train_masks]
        if (want == "auto" and getattr(self.search, "_lease", None) is not None and world == 1 and hasattr(ds, "covariance_count")
                and ds.covariance_count() >= self.n_splits):
            # (a LEASED dataset -- the lines of one LineSearchCV, same cv, same masks: an earlier line built these very Grams and
            #  the entries are found by their fingerprints.  Not for a dataset that merely comes out of the cache with Grams of
            #  some other search on it -- other splits: every fold would be built afresh, mask by mask -- and not among ranks,
            #  where the count is one rank's state and the build a collective: there the cost model below decides, from
            #  quantities that are the same on every rank)
            want = True
        if want == "auto":
            points = sum(len(idx) for call in calls for lane in call for _, idx in lane)
            reads = 1.0  # reads of X per pass
            if self.adaptive:
                # every cell is a loop of re-weighted solves, each about four passes, and their warm starts take their
                # residuals from X too: two reads per pass (measured, 60 cells x 5 rounds at 100 000 x 5 000: 0.186 s of
                # solves over X, 0.061 s from the Grams -- tools/adaptive_grid_big.py)
                points *= 4 * max(1, int(getattr(self.est, "max_iter", 1)))
                reads = 2.0
            elif world > 1:
                reads = 1.5  # pieces of paths append k times the columns per pass: 1.18 ms against 0.90 ms
            lease = getattr(self.search, "_lease", None)
            if lease is not None:  # (a line search: the Grams serve the lines still to come)
                points *= max(1, lease.repeats)
            passes = 1.1 * points / max(self.lanes, 1)
            saved = passes * (reads * 8.0 * n * p / 6.5e12 - 8.0 * p * p / 5.0e12)
            tests = getattr(self, "test_masks", None)
            partition = bool(tests) and len(tests) == self.n_splits and bool(np.all(np.sum(tests, axis=0) == 1.0))
            triangle = 1.3 * n * p * p / 50e12  # X^T X on the matrix cores (its lower triangle), set-up included
            grams = triangle if partition else triangle * (1.0 + float(np.mean([1.0 - np.mean(m) for m in self.train_masks])) * self.n_splits)
            if shared and partition:
                # a world-th of the products, plus the exchange: n_splits packed triangles at ~150 GB/s per rank
                grams = triangle / world + self.n_splits * 4.0 * p * p / 150e9
            if n * p < (1 << 26) or saved < 1.15 * grams:
                return False
        try:
            if len(self.train_masks) <= _engine.MAX_LANES and hasattr(ds, "covariance_folds"):
                ds.covariance_folds(self.train_masks, n_effs)
            else:
                for m in self.train_masks:
                    ds.covariance(m, int(m.sum()))
        except NotImplementedError:  # (a matter of the shape: every rank gets the same answer, before any collective)
            return False
        except MemoryError:
            if shared:
                raise  # (the other ranks are inside the collective: nothing to fall back to on one rank's say-so)
            return False
        return True

    def solve_share(self, ds, rank, world):
        """(cells of rank `rank`, number of solves that stopped short of the tolerance)"""
        calls = self.plan(world)[rank]
        local = {}
        largest = max(self.plan(world), key=lambda cs: sum(len(idx) for call in cs for lane in call for _, idx in lane))
        if self.covariance(ds, largest, world):
            self.opts["flags"] = self.opts.get("flags", 0) | _engine.FLAG_COVARIANCE
            # covariance passes serve sixteen lanes a call (a search over X may have been planned on thirty-two: the two halves
            # of the split pass on one read of X): the same decision on every rank, so every rank plans again alike
            cap = D.min_over_ranks(ds.max_lanes(self.opts["flags"]))
            if self.lanes > cap:
                self.lanes = cap
                self._plan_world = None
                calls = self.plan(world)[rank]
        run = self._run_adaptive if self.adaptive else self._run_call
        unconverged = self.search._run_batches(ds, calls, lambda d, call: run(d, call, local), self)
        return local, unconverged

    def _holdout_sse(self, ds, betas, f):
        """Squared hold-out error of the coefficient vectors `betas` (rows; with the intercept's coefficient last when the
        device copy carries the column of ones) on fold f's test rows.  On the device for matrices of any size
        (`slm_eval_sse_sparse` / `slm_eval_sse` with the test mask) -- except the reference's own sizes, where a device call
        (0.09 ms) costs more than the product on the host: there `X[test] @ betas.T` in numpy."""
        n, p = self.X.shape
        if n * (p + 16) > 131072:
            return ds.eval_sse(betas, self.test_masks[f])
        cache = self.__dict__.setdefault("_test_rows", {})
        if f not in cache:
            test = self.splits[f][1]
            cache[f] = (np.ascontiguousarray(self.X[test]), self.y[test])
        Xt, yt = cache[f]
        B = np.atleast_2d(betas)
        pred = Xt @ B[:, :p].T
        if self.intercept:
            pred = pred + B[:, p][None, :]
        r = pred - yt[:, None]
        return np.einsum("ij,ij->j", r, r)

    def _run_call(self, ds, call, local):
        """One call of the engine for the lanes of `call` on dataset `ds` (this stream's copy); scores go into `local`."""
        search, cands = self.search, self.candidates
        specs = []
        for lane in call:
            c0, f = self.units[lane[0][0]]
            vecs, _, _ = self.penalty_of(c0)
            segs = []
            for u, idx in lane:
                c, _ = self.units[u]
                _, (sa, sb, sd), _ = self.penalty_of(c)
                alphas = np.array([cands[self.combos[c][i]]["alpha"] for i in idx], dtype=float)
                segs.append(np.c_[alphas * sa, alphas * sb, np.full(len(alphas), sd)])
            pts, gam = _engine.lane_points(segs)
            specs.append(dict(points=pts, extrap=gam, a=vecs[0], b=vecs[1], d=vecs[2], row_weight=self.train_masks[f],
                              n_eff=len(self.splits[f][0])))
        n_fits = sum(len(idx) for lane in call for _, idx in lane)
        t_call = time.perf_counter()
        try:
            results = _solve_lanes_with_fallback(ds, specs, self.opts)
        except _engine.NonFiniteError:  # the counterpart of a failing fit in _fit_and_score
            if search.error_score == "raise":
                raise
            for lane in call:
                for u, idx in lane:
                    c, f = self.units[u]
                    local[(u, idx[0])] = ([self.combos[c][i] for i in idx], f, np.full(len(idx), search.error_score, dtype=float), 0.0)
            return 0
        dt = (time.perf_counter() - t_call) / max(1, n_fits)
        # hold-out scores, fold by fold: one scoring call for all the coefficient vectors of the call that share a test mask
        by_fold = defaultdict(list)
        for lane, res in zip(call, results):
            at = 0
            for u, idx in lane:
                by_fold[self.units[u][1]].append((u, idx, res.betas[at : at + len(idx)]))
                at += len(idx)
        for f, parts in by_fold.items():
            sse = self._holdout_sse(ds, np.vstack([b for _, _, b in parts]), f)
            sc = search._score_from_sse(sse, self.y[self.splits[f][1]])
            at = 0
            for u, idx, _ in parts:
                c = self.units[u][0]
                local[(u, idx[0])] = ([self.combos[c][i] for i in idx], f, sc[at : at + len(idx)], dt)
                at += len(idx)
        return sum(not r.converged for r in results)

    def _run_adaptive(self, ds, call, local):
        search = self.search
        batch = [self.units[lane[0][0]] for lane in call]
        us = [lane[0][0] for lane in call]
        # (one estimator per candidate, not per cell: the loops only read it -- a clone + set_params is 30 us, 50 cells of it
        #  1.5 ms of a 10 ms search)
        made = self.__dict__.setdefault("_cell_estimators", {})
        ests = []
        for c, _ in batch:
            ci = self.combos[c][0]
            if ci not in made:
                made[ci] = clone(self.est).set_params(**self.candidates[ci])
            ests.append(made[ci])
        t_batch = time.perf_counter()
        try:
            fits = _adaptive_lanes(ds, ests, self.X, [self.train_masks[f] for _, f in batch],
                                   [len(self.splits[f][0]) for _, f in batch], self.opts, self.with_intercept, want_weights=False)
        except _engine.NonFiniteError:
            if search.error_score == "raise":
                raise
            for u, (c, f) in zip(us, batch):
                local[(u, 0)] = (self.combos[c], f, np.full(len(self.combos[c]), search.error_score, dtype=float), 0.0)
            return 0
        dt = (time.perf_counter() - t_batch) / len(batch)
        # hold-out scores fold by fold: one scoring call for all the cells of the call that share a test mask
        by_fold = defaultdict(list)
        for k, (_, f) in enumerate(batch):
            by_fold[f].append(k)
        for f, ks in by_fold.items():
            sse = self._holdout_sse(ds, np.vstack([fits[k]["beta"] for k in ks]), f)
            sc = search._score_from_sse(sse, self.y[self.splits[f][1]])
            for k, one in zip(ks, np.atleast_1d(sc)):
                local[(us[k], 0)] = (self.combos[batch[k][0]], f, np.array([one]), dt)
        return sum(not i["converged"] for fit in fits for i in fit["infos"])

    def merge(self, cells):
        scores = np.full((len(self.candidates), self.n_splits), np.nan)
        fit_time = np.zeros((len(self.candidates), self.n_splits))
        for cis, f, sc, dt in cells.values():
            scores[cis, f] = sc
            fit_time[cis, f] = dt
        return scores, fit_time

    def refit(self, ds, best_params):
        search, p = self.search, self.X.shape[1]
        best = clone(self.est).set_params(**best_params)
        if self.adaptive:
            fit = _adaptive_lanes(ds, [best], self.X, [None], [self.X.shape[0]], self.opts, self.with_intercept)[0]
            beta_aug = fit["beta"]
            best.n_iter_ = fit["n_iter"]
            best.adaptive_weights_ = fit["weights"]
            best.solver_info_ = {"solves": fit["infos"]}
            if not all(i["converged"] for i in fit["infos"]):
                search._warn_unconverged(1, "the refit", self.opts)
        else:
            a, b, d, _, G_b = best._penalty(self.X)
            a, b, d = self.with_intercept(a, b, d, G_b if G_b is not None else p)
            res = ds.solve_path(
                [(1.0, 1.0, 1.0)],
                a=np.zeros(ds.p) if a is None else a,
                b=np.zeros(ds.n_groups) if b is None else b,
                d=np.zeros(ds.n_groups) if d is None else d,
                **self.opts,
            )
            beta_aug = res.betas[0].copy()
            best.solver_info_ = {"n_iter": int(res.n_iter[0]), "converged": res.converged}
            if not res.converged:
                search._warn_unconverged(1, "the refit", self.opts)
        best.coef_ = beta_aug[:p].copy()
        best.intercept_ = float(beta_aug[p]) if self.intercept else 0.0
        best.n_features_in_ = p
        return best


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
        lease = _DatasetLease()
        try:
            self._fit_lines(X, y, groups, fit_params, n_iter, n_params, methods, history, best, lease)
        finally:
            lease.close()
        self.history_ = history
        for attr in (v for v in vars(history[-1]) if v.endswith("_") and not v.startswith("__")):
            setattr(self, attr, getattr(history[-1], attr))
        return self

    def _fit_lines(self, X, y, groups, fit_params, n_iter, n_params, methods, history, best, lease):
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
            lease.repeats = n_iter - i  # the lines still to come on this dataset, this one included
            search._lease = lease
            try:
                search.fit(X, y, groups=groups, **fit_params)
            finally:
                del search._lease  # (a private attribute: not a constructor parameter, gone before anyone clones the search)
            best = deepcopy(search.best_params_)
            history.append(search)

    def _run_search(self, evaluate_candidates):
        """Unused: every line is its own GridSearchCV."""
        return


def _value_key(v):
    """What two grid values must share to count as the same value: the CONTENT of arrays (`repr` elides the middle of an
    array of more than a thousand entries -- two `group_weights` candidates that differ only there would be one unit,
    solved and scored with the first one's penalty), `repr` for everything else."""
    if isinstance(v, (np.ndarray, list, tuple)):
        try:
            a = np.asarray(v)
            if a.dtype != object:
                return ("array", a.dtype.str, a.shape, a.tobytes())
        except (ValueError, TypeError):
            pass
    return repr(v)


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
    out["flags"] = _backend.solve_flags(o)
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


def _adaptive_lanes_on_chip(ds, ests, st, p, row_weights, n_effs, opts, with_intercept, want_weights):
    """The loops of ``_adaptive_lanes`` inside one launch (``Dataset.solve_lanes_reweighted``): for reference-sized
    problems and the default update function -- every round used to be a call of its own, 0.5 ms of launch, wait and
    numpy per round of a 5 ms search.  ``None`` when this is not such a case (the rule is a user's function, the
    estimators differ in ``warm_start``, the problem is not one the on-chip solver takes, a round did not settle there):
    the caller's loop runs as before."""
    if not hasattr(ds, "solve_lanes_reweighted") or os.environ.get("SLM_HOST_ROUNDS"):
        return None
    rules = [e._reweight_rule(p, s["G"]) for e, s in zip(ests, st)]
    if any(r is None for r in rules) or len({bool(e.warm_start) for e in ests}) != 1:
        return None
    flags = int(opts.get("flags", 0))
    if not ests[0].warm_start:
        flags |= _engine.FLAG_COLD_START
    specs = []
    for e, s, rule, rw, ne in zip(ests, st, rules, row_weights, n_effs):
        a, b, d = with_intercept(*e._weights_to_penalty(s["w"], p, s["G"]), s["G"])
        specs.append(dict(points=np.ones((int(e.max_iter), 3)), a=a, b=b, d=d, row_weight=rw, n_eff=ne, reweight=rule))
    try:
        results, rounds = ds.solve_lanes_reweighted(specs, tol=opts.get("tol", 1e-8), max_iter=opts.get("max_iter", 10000), flags=flags)
    except NotImplementedError:
        return None
    fits = []
    for e, s, res, r in zip(ests, st, results, rounds):
        beta = res.betas[r - 1].copy()
        fit = dict(beta=beta, n_iter=r, infos=[{"n_iter": int(res.n_iter[k]), "converged": True} for k in range(r)])
        if want_weights:
            fit["weights"] = e._updated_weights(beta[:p], res.group_norms[r - 1][: s["G"]])
        fits.append(fit)
    return fits


def _adaptive_lanes(ds, ests, X, row_weights, n_effs, opts, with_intercept, want_weights=True):
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
    on_chip = _adaptive_lanes_on_chip(ds, ests, st, p, row_weights, n_effs, opts, with_intercept, want_weights)
    if on_chip is not None:
        return on_chip
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
