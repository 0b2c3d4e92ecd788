"""ctypes access to the plain-C oracle twin (``fista_ref.c``).  TEST INFRASTRUCTURE ONLY."""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .build import OUT, build

_lib = None


def _load():
    global _lib
    if _lib is None:
        path = OUT if os.path.exists(OUT) else build()
        lib = C.CDLL(path)
        vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
        lib.oracle_gradient.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp]
        lib.oracle_gradient.restype = dbl
        lib.oracle_fista.argtypes = [vp, i64, i64, i64, vp, vp, vp, vp, vp, vp, i32, dbl, dbl, i64, C.c_int, vp]
        lib.oracle_fista.restype = i64
        lib.oracle_numa_copy.argtypes = [vp, i64, i64]
        lib.oracle_numa_copy.restype = vp
        lib.oracle_free.argtypes = [vp]
        lib.oracle_free.restype = None
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def gradient(X, y, z, w=None):
    """(g, loss) with g = X^T (w .* (X z - y)) / n, single fused pass, all host cores (OpenMP)."""
    lib = _load()
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.float64)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    n, p = X.shape
    g = np.empty(p)
    loss = lib.oracle_gradient(_p(X), n, p, p, _p(y), _p(w), _p(z), _p(g))
    return g, loss


def fista(X, y, a, b, d, gidx, n_groups, beta0=None, L=None, tol=1e-13, max_iter=200000, restart=True, w=None):
    """Same contract as oracle.fista.fista, executed by the C twin.  Returns (beta, n_iter signed)."""
    from .fista import lipschitz

    lib = _load()
    X = np.ascontiguousarray(X, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    n, p = X.shape
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (p,)))
    b = np.ascontiguousarray(np.broadcast_to(np.asarray(b, dtype=np.float64), (n_groups,)))
    d = np.ascontiguousarray(np.broadcast_to(np.asarray(d, dtype=np.float64), (n_groups,)))
    gi = np.ascontiguousarray(gidx, dtype=np.int32)
    w = None if w is None else np.ascontiguousarray(w, dtype=np.float64)
    if L is None:
        L = lipschitz(X)
    beta = np.zeros(p) if beta0 is None else np.array(beta0, dtype=np.float64)
    it = lib.oracle_fista(
        _p(X), n, p, p, _p(y), _p(w), _p(a), _p(b), _p(d), _p(gi), n_groups, float(L), float(tol), int(max_iter),
        1 if restart else 0, _p(beta),
    )
    return beta, int(it)


class NumaMatrix:
    """Row-major copy of X whose pages were first touched in parallel (see oracle_numa_copy); use as
    ``with NumaMatrix(X) as Xn:`` and pass ``Xn`` wherever these wrappers take ``X``."""

    def __init__(self, X):
        lib = _load()
        X = np.ascontiguousarray(X, dtype=np.float64)
        self.shape = X.shape
        self._ptr = lib.oracle_numa_copy(_p(X), X.shape[0], X.shape[1])
        if not self._ptr:
            raise MemoryError("oracle_numa_copy failed")
        buf = (C.c_double * (X.shape[0] * X.shape[1])).from_address(self._ptr)
        self.array = np.frombuffer(buf, dtype=np.float64).reshape(X.shape)

    def __enter__(self):
        return self.array

    def __exit__(self, *exc):
        self.array = None
        _load().oracle_free(self._ptr)
        self._ptr = None


def num_threads() -> int:
    return int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
