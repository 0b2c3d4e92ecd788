"""The solve seam between the estimator surface and the HIP engine.

``solve(problem)`` is the counterpart of ``CVXRegressor._solve`` (reference
src/sparselm/model/_base.py:512-519): it receives the already preprocessed ``(X, y)`` and the
penalty ``(a, b, d)`` on groups ``gidx`` and returns the minimiser.  The only backend shipped is
the HIP engine; ``use_backend`` exists so tests can time or check the surface against another
implementation without touching product code.
"""

from __future__ import annotations

import os
import warnings
from contextlib import contextmanager

import numpy as np

from . import _engine

_KNOWN_OPTIONS = {"tol", "max_iter", "L", "restart", "check_every", "device", "on_chip", "covariance"}


class DatasetCache:
    """Device-resident datasets kept across ``fit`` calls -- the counterpart of the reference's ``cached_X_`` /
    ``cached_y_`` (src/sparselm/model/_base.py:182-195: a re-fit on identical data re-uses the cvxpy problem).

    scikit-learn's stock ``GridSearchCV`` fits ``clone(estimator)`` on ``X[train]`` once per (candidate, fold):
    fifty fits of the README example touch five distinct training sets.  Creating an engine dataset costs a few
    dozen device allocations (about a millisecond: more than a small solve), so datasets are kept in a small
    LRU keyed by the CONTENT of what was uploaded -- a 128-bit xxh3 digest of X, y and the row weights, with
    shape, layout, the centring flag, device and process id.  A key is only computed for arrays up to
    ``max_bytes`` (hashing a 4 GB matrix costs more than uploading it); larger ones are not cached.  A dataset
    handed out is marked busy until ``release``: a second user in another thread gets a fresh one.
    """

    def __init__(self, capacity=8, max_bytes=64 << 20):
        import threading

        self.capacity, self.max_bytes = int(capacity), int(max_bytes)
        self._lock = threading.Lock()
        self._items = {}  # key -> [dataset, x_mean, y_mean, busy, stamp]
        self._clock = 0
        self.hits = self.misses = 0

    def _key(self, engine, X, y, row_weight, center):
        import os

        try:
            import xxhash

            new_digest = xxhash.xxh3_128
        except ImportError:  # (xxhash is optional: the standard library's blake2b, a few times slower, serves as well)
            import hashlib

            def new_digest():
                return hashlib.blake2b(digest_size=16)

        X = np.asarray(X)
        if X.nbytes > self.max_bytes or X.ndim != 2:
            return None
        X = np.asarray(X, dtype=np.float64)
        fortran = X.flags.f_contiguous and not X.flags.c_contiguous
        h = new_digest()
        h.update(np.ascontiguousarray(X.T if fortran else X))  # (a view when the layout already fits: no copy)
        h.update(np.ascontiguousarray(y, dtype=np.float64))
        if row_weight is not None:
            h.update(np.ascontiguousarray(row_weight, dtype=np.float64))
        return (os.getpid(), engine.device_id, X.shape, fortran, row_weight is not None, bool(center), h.digest())

    def acquire(self, engine, X, y, row_weight, center, check_finite=False):
        """(dataset, x_mean, y_mean, key): from the cache when the same content was uploaded before.  ``check_finite``: the
        caller skipped the host-side scan of X (``fit`` on large arrays): it is made on the device copy, before anything is
        done to it, and raises scikit-learn's ValueError."""
        key = self._key(engine, X, y, row_weight, center)
        if key is not None:
            hit = None
            with self._lock:
                item = self._items.get(key)
                if item is not None and not item[3] and getattr(item[0], "_h", None):
                    item[3] = True
                    self._clock += 1
                    item[4] = self._clock
                    self.hits += 1
                    hit = item
            if hit is not None:
                if check_finite:  # (uploaded by a caller that had not asked: the scan -- a kernel, a copy, a wait -- runs outside the lock)
                    ds = hit[0]
                    try:
                        raise_if_nonfinite(ds)  # scikit-learn's messages; closes the dataset on the way out
                    except ValueError:
                        with self._lock:  # a design that is not finite does not stay in the cache
                            if self._items.get(key) is hit:
                                del self._items[key]
                        raise
                return hit[0], hit[1], hit[2], key
        ds = engine.dataset(X, y, row_weight=row_weight)
        if check_finite:
            raise_if_nonfinite(ds)
        # fit_intercept: centre the device copy in place (no centred host copy of X is ever made)
        x_mean, y_mean = ds.center() if center else (None, None)
        self.misses += 1
        return ds, x_mean, y_mean, key

    def release(self, ds, x_mean, y_mean, key):
        if key is None:
            ds.close()
            return
        evict = []
        with self._lock:
            item = self._items.get(key)
            if item is not None and item[0] is ds:
                item[3] = False
            elif item is None or not getattr(item[0], "_h", None):  # (a closed dataset -- its engine went -- gives way)
                self._clock += 1
                self._items[key] = [ds, x_mean, y_mean, False, self._clock]
                while len(self._items) > self.capacity:
                    idle = [(v[4], k) for k, v in self._items.items() if not v[3]]
                    if not idle:
                        break
                    evict.append(self._items.pop(min(idle)[1])[0])
            else:  # the same content is cached already (another thread got there first)
                evict.append(ds)
        for d in evict:
            d.close()

    def discard(self, ds, key):
        """A dataset handed out by `acquire` that turned out unusable: out of the cache, closed."""
        with self._lock:
            item = self._items.get(key) if key is not None else None
            if item is not None and item[0] is ds:
                del self._items[key]
        ds.close()

    def clear(self):
        import os

        with self._lock:
            items, self._items = list(self._items.items()), {}
        for key, item in items:
            if key[0] == os.getpid():  # (a forked child never touches its parent's handles)
                item[0].close()


_dataset_cache = DatasetCache()
# (device memory is handed back while the HIP runtime is still up: interpreter teardown destroys objects in no
#  particular order, and a dataset released after the runtime's own static destructors aborts the process)
import atexit  # noqa: E402

atexit.register(_dataset_cache.clear)


def dataset_cache() -> DatasetCache:
    return _dataset_cache


def raise_if_nonfinite(ds):
    """scikit-learn's error for a design with a NaN or an infinity (``sklearn.utils.validation._assert_all_finite``), from a
    scan of the device copy; the dataset is closed on the way out."""
    kind = ds.nonfinite()
    if kind:
        ds.close()
        if kind & 1:
            raise ValueError("Input X contains NaN.")
        raise ValueError("Input X contains infinity or a value too large for dtype('float64').")


class SolveProblem:
    """Device-resident problem: upload once, solve many penalties (adaptive loops, paths); the dataset itself
    outlives the problem in the ``DatasetCache``."""

    def __init__(self, backend, X, y, gidx, n_groups, options, row_weight=None, center=False, cache=True, check_finite=False):
        self.backend = backend
        self.options = options
        self.p = X.shape[1]
        self.n_groups = n_groups
        eng = _engine.get_engine(options.get("device"))
        if cache:
            self.ds, self.x_mean, self.y_mean, self._key = _dataset_cache.acquire(eng, X, y, row_weight, center, check_finite)
        else:  # a dataset of its own: the caller is going to change its targets (set_targets)
            self.ds = eng.dataset(X, y, row_weight=row_weight)
            if check_finite:
                raise_if_nonfinite(self.ds)
            self.x_mean, self.y_mean = self.ds.center() if center else (None, None)
            self._key = None
        self._private = not cache
        # (a cached dataset may carry another estimator's groups: always set them)
        try:
            self.ds.set_groups(gidx, n_groups if gidx is not None else None)
        except BaseException:  # (never leave a cache entry marked busy behind)
            _dataset_cache.discard(self.ds, self._key)
            self.ds = None
            raise

    def solve(self, a, b, d, beta0=None, want_group_norms=False):
        """One minimisation with penalty (a, b, d); returns (beta, group_norms or None, info)."""
        o = self.options
        flags = solve_flags(o)
        if o.get("covariance") is True:
            # asked for by name (a single fit never takes it by itself: the Gram of all rows costs about eighty passes
            # over X): worth it for re-weighting loops with many rounds, refits on a cached dataset, dense solutions
            try:
                self.ds.covariance(None, 0)  # (found again in microseconds once built)
                flags |= _engine.FLAG_COVARIANCE
            except NotImplementedError:
                pass
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
        # (copies: an estimator's coef_ must not keep a block of the engine's page-locked result pool alive)
        gn = None if res.group_norms is None else res.group_norms[0].copy()
        return res.betas[0].copy(), gn, info

    def solve_rounds(self, a, b, d, rule, rounds, beta0=None, cold=False):
        """The re-weighting rounds of an Adaptive* fit in ONE launch (``Dataset.solve_lanes_reweighted``; ``rule`` as
        ``AdaptiveLasso._reweight_rule`` gives it): ``(beta, group_norms, infos)`` of the last round run, one info per
        round -- or ``None`` where that does not apply (not a problem for the on-chip solver, a round that did not settle
        there, ``covariance=True``): the caller then loops over ``solve``."""
        o = self.options
        flags = solve_flags(o)
        if not (flags & _engine.FLAG_ON_CHIP) or o.get("covariance") is True or os.environ.get("SLM_HOST_ROUNDS"):
            return None
        if cold:
            flags |= _engine.FLAG_COLD_START
        try:
            (res,), (r,) = self.ds.solve_lanes_reweighted(
                [dict(points=np.ones((int(rounds), 3)), a=a, b=b, d=d, beta0=beta0, reweight=rule)],
                tol=float(o.get("tol", default_tol(self.ds.n, self.ds.p))), max_iter=int(o.get("max_iter", 10000)), flags=flags)
        except NotImplementedError:
            return None
        infos = [{"n_iter": int(res.n_iter[k]), "converged": True, "resid": float(res.resid[k]), "L": float(res._infos["L"][k]),
                  "loss": float(res.loss[k]), "wall_ms": res.wall_ms} for k in range(r)]
        return res.betas[r - 1].copy(), res.group_norms[r - 1].copy(), infos

    def set_targets(self, y):
        """New targets on the same design (problems opened with ``cache=False`` only: a cached dataset is
        found again by the content it was uploaded with)."""
        if not self._private:
            raise RuntimeError("set_targets needs a problem opened with cache=False")
        self.ds.set_targets(y)

    def close(self):
        if self.ds is not None:
            _dataset_cache.release(self.ds, self.x_mean, self.y_mean, self._key)
            self.ds = None


class HipBackend:
    name = "hip"
    # sample weights and centring are applied on the device (row weights in the fused kernel,
    # slm_dataset_center): the estimator hands over the raw validated arrays
    native_preprocessing = True

    def problem(self, X, y, gidx, n_groups, options, row_weight=None, center=False, cache=True, check_finite=False) -> SolveProblem:
        return SolveProblem(self, X, y, gidx, n_groups, options, row_weight=row_weight, center=center, cache=cache,
                            check_finite=check_finite)


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


def solve_flags(options) -> int:
    """Engine flags of a fit: ``restart=False`` is FISTA without the momentum restart; the on-chip solver for problems
    that fit a workgroup (the reference's own sizes: one launch per call instead of a dozen per pass) unless
    ``on_chip=False``."""
    flags = 0 if options.get("restart", True) else _engine.FLAG_NO_RESTART
    if options.get("on_chip", True):
        flags |= _engine.FLAG_ON_CHIP
    return flags


def default_tol(n: int, p: int) -> float:
    """Stopping tolerance when ``solver_options`` names none.  A point is accepted when its KKT residual (the
    prox-gradient mapping) is below ``tol * mu * ||beta||``, mu the strong-convexity estimate of the face
    (include/slm_engine.h, ``slm_solve_opts.tol``): ``tol`` bounds the relative distance to the
    minimiser whatever the conditioning.  Small problems -- the reference's own sizes -- are launch-bound, so two
    more digits cost little there: 1e-10 below 2^26 matrix entries, 1e-8 (5e-11...2e-8 from scikit-learn's
    coordinate descent on the BASELINE designs) above."""
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
