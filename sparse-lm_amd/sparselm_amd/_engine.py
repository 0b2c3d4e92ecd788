"""ctypes binding of the C ABI in ``include/slm_engine.h`` (``libslm_hip.so``).

This is the only place Python touches the HIP engine.  There is no CPU fallback: if the shared
library is missing, or no gfx950 device is visible, every entry point raises.  The GIL is released
for the duration of each foreign call (ctypes does that for ``CDLL`` functions).

The boundary mirrors ``CVXRegressor._solve(X, y, solver_options) -> beta`` of the reference
(src/sparselm/model/_base.py:512-519): ``Dataset`` holds the preprocessed ``(X, y)`` in HBM and
``Dataset.solve_path`` returns the minimiser(s).
"""

from __future__ import annotations

import ctypes as C
import os
import sys
import threading
import weakref

import numpy as np

_LIB_NAME = "libslm_hip.so"
_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib")

SLM_OK = 0
SLM_ERR_BAD_ARG = 1
SLM_ERR_OOM = 2
SLM_ERR_HIP = 3
SLM_ERR_NO_DEVICE = 4
SLM_ERR_COMM = 5
SLM_ERR_NOT_CONVERGED = 6
SLM_ERR_NON_FINITE = 7
SLM_ERR_UNSUPPORTED = 8

FLAG_NO_RESTART = 1
FLAG_PROFILE = 2
FLAG_COLD_START = 4
FLAG_FRESH_L = 8
FLAG_FISTA_ONLY = 16
FLAG_WORKING_SET = 32  # force the Gram-assisted refinement even for small X
FLAG_NO_WORKING_SET = 64
FLAG_ON_CHIP = 128  # problems whose Gram matrix fits a workgroup: one launch per call (csrc/small_kernels.hpp)
FLAG_COVARIANCE = 256  # passes from the Grams of the call's row sets (Dataset.covariance; csrc/cov_kernels.hpp)
FLAG_PROFILE_UNIT = 1024  # with FLAG_PROFILE: the event bracket spans residuals + X^T R, the whole gradient unit of the split pass
FLAG_NO_MODEL_GRAM = 512  # lanes beyond the working set's 512 columns take plain steps, no rounds on the model Gram (csrc/mg_kernels.hpp)

COMM_ID_BYTES = 128
ABI_VERSION = 19  # SLM_ABI_VERSION of include/slm_engine.h this binding was written against

# every symbol include/slm_engine.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = (
    "slm_abi_version",
    "slm_last_error",
    "slm_device_count",
    "slm_host_alloc",
    "slm_host_free",
    "slm_engine_create",
    "slm_engine_destroy",
    "slm_engine_synchronize",
    "slm_reload_knobs",
    "slm_engine_device_info",
    "slm_dataset_create",
    "slm_dataset_create_device",
    "slm_dataset_create_synthetic",
    "slm_dataset_nonfinite",
    "slm_dataset_destroy",
    "slm_dataset_shape",
    "slm_dataset_download",
    "slm_dataset_center",
    "slm_dataset_set_row_weights",
    "slm_dataset_set_targets",
    "slm_dataset_clone",
    "slm_dataset_set_groups",
    "slm_dataset_lipschitz",
    "slm_dataset_max_lanes",
    "slm_dataset_path_lanes",
    "slm_gradient",
    "slm_gradient_ex",
    "slm_eval_sse",
    "slm_eval_sse_sparse",
    "slm_dense_spd_solve",
    "slm_solve_path",
    "slm_solve_lanes",
    "slm_solve_lanes_reweighted",
    "slm_solve_path_lanes",
    "slm_solve_standardized_sgl",
    "slm_dataset_covariance",
    "slm_dataset_covariance_folds",
    "slm_dataset_covariance_folds_begin",
    "slm_dataset_covariance_folds_finish",
    "slm_dataset_covariance_count",
    "slm_dataset_covariance_clear",
    "slm_dataset_covariance_download",
    "slm_dataset_model_gram",
    "slm_dataset_read_ceiling",
    "slm_dataset_set_replicated",
    "slm_comm_unique_id",
    "slm_comm_init",
    "slm_comm_info",
    "slm_comm_collectives",
    "slm_comm_all_reduce_probe",
    "slm_comm_init_local",
    "slm_dataset_set_global_rows",
    "slm_comm_destroy",
)


class EngineError(RuntimeError):
    """HIP / RCCL / device failure reported by the engine."""


class NonFiniteError(EngineError):
    """The iterate became non-finite (counterpart of cvxpy's SolverError / 'infeasible')."""


class _PenaltyStruct(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("d", C.c_void_p)]


class _GradientOpts(C.Structure):
    _fields_ = [("route", C.c_int32), ("n_lanes", C.c_int32), ("lane_out", C.c_int32), ("probe_lanes", C.c_int32),
                ("xtr_only", C.c_int32)]


class _PathPoint(C.Structure):
    _fields_ = [("sa", C.c_double), ("sb", C.c_double), ("sd", C.c_double), ("extrap", C.c_double)]


class _SolveOpts(C.Structure):
    _fields_ = [
        ("tol", C.c_double),
        ("max_iter", C.c_int32),
        ("check_every", C.c_int32),
        ("L", C.c_double),
        ("flags", C.c_uint32),
    ]


class _PointInfo(C.Structure):
    _fields_ = [
        ("n_iter", C.c_int32),
        ("status", C.c_int32),
        ("resid", C.c_double),
        ("beta_norm", C.c_double),
        ("loss", C.c_double),
        ("L", C.c_double),
        ("mode", C.c_int32),
        ("rejects", C.c_int32),
        ("kkt", C.c_double),
        ("mu", C.c_double),
    ]


class _SolveStats(C.Structure):
    _fields_ = [
        ("grad_launches", C.c_int64),
        ("grad_timed", C.c_int64),
        ("grad_ms_total", C.c_double),
        ("wall_ms", C.c_double),
        ("lipschitz_ms", C.c_double),
        ("ws_builds", C.c_int64),
        ("ws_appends", C.c_int64),
        ("ws_refined", C.c_int64),
        ("ws_misses", C.c_int64),
        ("ws_columns", C.c_int64),
        ("ws_inner_iters", C.c_int64),
        ("ws_direct_steps", C.c_int64),
        ("mg_rounds", C.c_int64),
        ("mg_inner_iters", C.c_int64),
        ("mg_rejected", C.c_int64),
        ("mg_build_ms", C.c_double),
        ("light_passes", C.c_int64),
        ("light_columns", C.c_int64),
    ]


class _Lane(C.Structure):
    _fields_ = [
        ("pen", C.POINTER(_PenaltyStruct)),
        ("points", C.POINTER(_PathPoint)),
        ("n_points", C.c_int32),
        ("beta0", C.c_void_p),
        ("row_weight", C.c_void_p),
        ("n_eff", C.c_int64),
        ("betas_out", C.c_void_p),
        ("group_norms_out", C.c_void_p),
        ("infos", C.POINTER(_PointInfo)),
    ]


class _Reweight(C.Structure):
    _fields_ = [
        ("coef_scale", C.c_double),
        ("group_scale", C.c_void_p),
        ("numerator", C.c_double),
        ("eps", C.c_double),
        ("tol", C.c_double),
        ("n_coef", C.c_int32),
        ("n_group", C.c_int32),
    ]


MAX_LANES = 16  # lanes of a call of independent problems (a half of the split pass: the sixteen columns of an MFMA operand)
MAX_LANES_WIDE = 32  # SLM_MAX_LANES: lanes a shared path may run on -- two halves on one read of X (xtr32_mfma_kernel)
MAX_CELLS = 64  # SLM_MAX_CELLS: lanes of a call the on-chip solver takes (Dataset.max_lanes tells which limit applies)

# numpy views of the two per-point structs (no per-point Python objects on the way in or out)
_INFO_DTYPE = np.dtype(
    [("n_iter", "<i4"), ("status", "<i4"), ("resid", "<f8"), ("beta_norm", "<f8"), ("loss", "<f8"), ("L", "<f8"),
     ("mode", "<i4"), ("rejects", "<i4"), ("kkt", "<f8"), ("mu", "<f8")]
)
assert _INFO_DTYPE.itemsize == C.sizeof(_PointInfo) and C.sizeof(_PathPoint) == 32


def _points_block(pts, gam):
    """(K, 4) float64 rows (sa, sb, sd, extrap) == slm_path_point[K]"""
    out = np.empty((pts.shape[0], 4))
    out[:, :3] = pts
    out[:, 3] = gam
    return out


def _as(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))

WS_COLUMNS = 512  # WS_KCAP of the engine: columns a working set / a sparse scoring call can hold

_lib = None
_lib_lock = threading.Lock()


def library_path() -> str:
    return os.environ.get("SLM_HIP_LIBRARY", os.path.join(_LIB_DIR, _LIB_NAME))


def load_library():
    """dlopen the engine; raises ``EngineError`` with build instructions if it is missing."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            raise EngineError(
                f"{path} not found: build the HIP engine first "
                "(python -c 'import __graft_entry__ as g; g.build()' or "
                "`python sparse-lm_amd/build.py`). sparselm_amd has no CPU fallback."
            )
        try:
            lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError as exc:  # missing ROCm runtime etc.
            raise EngineError(f"cannot load {path}: {exc}") from exc
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        P = C.POINTER
        lib.slm_abi_version.restype = C.c_int
        lib.slm_last_error.restype = C.c_char_p
        if lib.slm_abi_version() != ABI_VERSION:
            raise EngineError(
                f"{path} has ABI version {lib.slm_abi_version()}, this binding needs {ABI_VERSION}: "
                "rebuild it with `python sparse-lm_amd/build.py --force`"
            )
        sigs = {
            "slm_device_count": [P(C.c_int)],
            "slm_engine_create": [C.c_int, P(vp)],
            "slm_engine_destroy": [vp],
            "slm_engine_synchronize": [vp],
            "slm_engine_device_info": [vp, P(i64), C.c_char_p, C.c_int],
            "slm_dataset_create": [vp, vp, i64, i64, i64, i64, vp, vp, P(vp)],
            "slm_dataset_create_device": [vp, vp, i64, i64, i64, vp, vp, P(vp)],
            "slm_dataset_create_synthetic": [vp, i64, i64, C.c_uint64, i64, vp, dbl, P(vp)],
            "slm_dataset_nonfinite": [vp, P(i32)],
            "slm_host_alloc": [C.c_size_t, P(vp)],
            "slm_host_free": [vp],
            "slm_dataset_destroy": [vp],
            "slm_dataset_shape": [vp, P(i64), P(i64), P(i64)],
            "slm_dataset_download": [vp, vp, vp],
            "slm_dataset_center": [vp, vp, P(dbl)],
            "slm_dataset_set_row_weights": [vp, vp],
            "slm_dataset_set_targets": [vp, vp],
            "slm_dataset_clone": [vp, vp, P(vp)],
            "slm_dataset_set_groups": [vp, vp, i32],
            "slm_dataset_lipschitz": [vp, P(dbl)],
            "slm_dataset_max_lanes": [vp, C.c_uint32, P(i32)],
            "slm_dataset_path_lanes": [vp, i32, C.c_uint32, P(i32)],
            "slm_gradient": [vp, vp, vp, P(dbl), i32, P(dbl)],
            "slm_gradient_ex": [vp, vp, P(_GradientOpts), vp, P(dbl), i32, P(dbl)],
            "slm_reload_knobs": [],
            "slm_eval_sse": [vp, vp, i32, vp, vp],
            "slm_eval_sse_sparse": [vp, vp, i32, vp, i32, vp, vp],
            "slm_dense_spd_solve": [vp, vp, i32, vp, vp, P(dbl)],
            "slm_solve_path": [
                vp,
                P(_PenaltyStruct),
                P(_PathPoint),
                i32,
                P(_SolveOpts),
                vp,
                vp,
                vp,
                P(_PointInfo),
                P(_SolveStats),
            ],
            "slm_solve_lanes": [vp, P(_Lane), i32, P(_SolveOpts), P(_SolveStats)],
            "slm_solve_lanes_reweighted": [vp, P(_Lane), P(_Reweight), i32, P(_SolveOpts), P(_SolveStats), P(i32)],
            "slm_solve_path_lanes": [
                vp, P(_PenaltyStruct), P(_PathPoint), i32, i32, P(_SolveOpts), vp, vp, vp, P(_PointInfo), P(_SolveStats),
            ],
            "slm_solve_standardized_sgl": [vp, vp, vp, P(_SolveOpts), dbl, i32, vp, i32, vp, vp, P(_PointInfo)],
            "slm_dataset_covariance": [vp, vp, i64],
            "slm_dataset_covariance_folds": [vp, vp, vp, i32],
            "slm_dataset_covariance_folds_begin": [vp, vp, vp, i32, P(i32)],
            "slm_dataset_covariance_folds_finish": [vp],
            "slm_dataset_covariance_count": [vp, P(i32)],
            "slm_dataset_covariance_clear": [vp],
            "slm_dataset_covariance_download": [vp, i32, vp, vp, vp],
            "slm_dataset_model_gram": [vp, vp],
            "slm_dataset_read_ceiling": [vp, i32, P(dbl), P(dbl)],
            "slm_dataset_set_replicated": [vp, i32],
            "slm_comm_unique_id": [vp],
            "slm_comm_init": [vp, i32, i32, vp],
            "slm_comm_info": [vp, P(i32), P(i32)],
            "slm_comm_collectives": [vp, P(i64)],
            "slm_comm_all_reduce_probe": [vp, i64, i32, P(dbl)],
            "slm_comm_init_local": [P(vp), i32, dbl],
            "slm_dataset_set_global_rows": [vp, i64],
            "slm_comm_destroy": [vp],
        }
        for name, argtypes in sigs.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int
        _lib = lib
        return lib


_binding = None  # the compiled binding of the hot calls (csrc/binding.cpp), False when it is not to be used


def load_binding():
    """The pybind11 module over the hot calls (``slm_solve_lanes``, ``slm_solve_path_lanes``, ``slm_dataset_create``), or
    None: built next to the engine by ``sparse-lm_amd/build.py``; ``SLM_NO_BINDING=1`` (A/B runs, the ABI tests) and a library
    taken from ``SLM_HIP_LIBRARY`` (the module is linked against the one in ``_lib``) leave every call to ctypes."""
    global _binding
    if _binding is None:
        lib = load_library()
        # (SLM_BINDING_PATH: another build of the same module -- tools/sanitize.sh loads the ASan + UBSan one)
        path = os.environ.get("SLM_BINDING_PATH") or os.path.join(_LIB_DIR, "_slm_binding.so")
        if os.environ.get("SLM_NO_BINDING") or os.environ.get("SLM_HIP_LIBRARY") or not os.path.exists(path):
            _binding = False
        else:
            import importlib.util

            spec = importlib.util.spec_from_file_location("_slm_binding", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            if mod.abi_version() != lib.slm_abi_version() or mod.info_record_bytes() != _INFO_DTYPE.itemsize:
                # a module left over from another ABI: an accelerator that does not fit is not used (ctypes serves every
                # call), and says so once
                import warnings

                warnings.warn(f"{path} does not match libslm_hip.so (rebuild with `python sparse-lm_amd/build.py --force`): "
                              "falling back on the ctypes binding", RuntimeWarning)
                _binding = False
            else:
                mod.set_error_types(EngineError, NonFiniteError)
                _binding = mod
    return _binding or None


def _check(rc: int):
    if rc == SLM_OK:
        return
    msg = (_lib.slm_last_error() or b"").decode("utf-8", "replace")
    if rc == SLM_ERR_BAD_ARG:
        raise ValueError(msg)
    if rc == SLM_ERR_OOM:
        raise MemoryError(msg)
    if rc == SLM_ERR_NON_FINITE:
        raise NonFiniteError(msg)
    if rc == SLM_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise EngineError(f"[slm status {rc}] {msg}")


def device_count() -> int:
    lib = load_library()
    n = C.c_int(0)
    rc = lib.slm_device_count(C.byref(n))
    return n.value if rc == SLM_OK else 0


def _f64(arr, name, shape=None):
    a = np.ascontiguousarray(arr, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"{name} has shape {a.shape}, expected {shape}")
    return a


def _ptr(a):
    # (the address as an int: what a c_void_p argument or field takes; `ctypes.data_as` costs twice as much, and a
    #  sixteen-lane call asks eighty times)
    return None if a is None else a.ctypes.data


_extrap_cache: dict = {}  # bytes of the points -> secant factors (the folds and rows of a grid walk the same path)


def path_extrapolation(pts) -> np.ndarray:
    """Secant factors gamma_k = (s_k - s_{k-1}) / (s_{k-1} - s_{k-2}) for a path whose points are all
    multiples s_k of one penalty direction (rank-one (K, 3) array); zeros otherwise."""
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
    key = pts.tobytes()
    hit = _extrap_cache.get(key)
    if hit is None:
        hit = _path_extrapolation(pts)
        if len(_extrap_cache) >= 4096:  # (a grid cut into pieces brings a few hundred distinct ones)
            _extrap_cache.clear()
        _extrap_cache[key] = hit
    return hit.copy()


def _path_extrapolation(pts) -> np.ndarray:
    K = pts.shape[0]
    gam = np.zeros(K)
    if K < 3:
        return gam
    ref = pts[np.argmax(np.abs(pts).sum(axis=1))]
    nrm = float(ref @ ref)
    if nrm <= 0.0:
        return gam
    s = pts @ ref / nrm
    # (plain array arithmetic: this runs between two solves, with the device idle -- a Python loop over the points
    #  and np.allclose cost 0.15 ms per 50-point path, 3 % of the solve itself)
    if np.max(np.abs(s[:, None] * ref[None, :] - pts)) > 1e-12 * np.max(np.abs(pts)):
        return gam  # the penalty changes shape along the path: prediction would be meaningless
    den = s[1:-1] - s[:-2]
    num = s[2:] - s[1:-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        g = num / den
    ok = (den != 0.0) & np.isfinite(g) & (np.abs(g) <= 10.0)
    gam[2:] = np.where(ok, g, 0.0)
    return gam


def lane_points(segments) -> tuple[np.ndarray, np.ndarray]:
    """(points, extrap) of a lane that walks several pieces of paths one after the other (the lanes
    ``distributed.plan_lane_calls`` plans): ``segments`` is a list of (K_s, 3) arrays in the order the lane takes
    them; the secant factors are those of each piece on its own, zero on the first two points of every piece (the
    two solutions before them belong to another path)."""
    pts = [np.ascontiguousarray(s, dtype=np.float64).reshape(-1, 3) for s in segments]
    return np.vstack(pts), np.concatenate([path_extrapolation(s) for s in pts])


_STATS_FIELDS = ("grad_launches", "grad_timed", "grad_ms_total", "wall_ms", "lipschitz_ms", "ws_builds", "ws_appends",
                 "ws_refined", "ws_misses", "ws_columns", "ws_inner_iters", "ws_direct_steps", "mg_rounds", "mg_inner_iters",
                 "mg_rejected", "mg_build_ms", "light_passes", "light_columns")


class PathResult:
    """The solutions of one lane's path.  ``betas`` (n_points, p) and ``group_norms`` (n_points, G) or None are views of the
    call's result blocks; the per-point records and the call's statistics are read where they are asked for (a
    sixteen-lane call used to build 160 small arrays nobody looked at):

    ``n_iter``, ``status`` (SLM_OK or SLM_ERR_NOT_CONVERGED), ``resid``, ``beta_norm``, ``loss``, ``mode`` (1 = spectral
    steps, 0 = FISTA, 2 = the on-chip solver), ``kkt`` (KKT residual at exit), ``mu`` (strong-convexity estimate the point
    was accepted with): arrays of n_points; ``L``: inverse step at the last point; ``converged``; and the statistics of the
    call, shared by its lanes: ``grad_launches``, ``grad_timed``, ``grad_ms_total``, ``wall_ms``, ``lipschitz_ms``,
    ``ws_builds`` (working sets selected from scratch; 0: refinement not used), ``ws_appends``, ``ws_refined``,
    ``ws_misses``, ``ws_columns`` (columns in the working set at the end), ``ws_inner_iters``, ``ws_direct_steps``,
    ``mg_rounds`` / ``mg_inner_iters`` / ``mg_rejected`` / ``mg_build_ms`` (the model Gram: rounds of lanes beyond the
    working set, their inner iterations, proposals the true objective rejected, build time inside the call),
    ``light_passes`` / ``light_columns`` (re-verifications after a miss that were certified partial passes instead of passes
    over X -- csrc/light_kernels.hpp -- and the borderline columns they read; not counted in ``grad_launches``)."""

    __slots__ = ("betas", "group_norms", "_infos", "_stats")

    def __init__(self, betas, group_norms, infos, stats):
        self.betas, self.group_norms, self._infos, self._stats = betas, group_norms, infos, stats

    n_iter = property(lambda self: self._infos["n_iter"].astype(np.int64))
    status = property(lambda self: self._infos["status"].astype(np.int64))
    resid = property(lambda self: self._infos["resid"].copy())
    beta_norm = property(lambda self: self._infos["beta_norm"].copy())
    loss = property(lambda self: self._infos["loss"].copy())
    mode = property(lambda self: self._infos["mode"].astype(np.int64))
    kkt = property(lambda self: self._infos["kkt"].copy())
    mu = property(lambda self: self._infos["mu"].copy())
    L = property(lambda self: float(self._infos["L"][-1]))

    @property
    def converged(self) -> bool:
        return not self._infos["status"].any()  # (SLM_OK == 0)

    def __getattr__(self, name):  # the call's statistics
        try:
            v = self._stats[_STATS_FIELDS.index(name)]
        except ValueError:
            raise AttributeError(name) from None
        return float(v) if name in ("grad_ms_total", "wall_ms", "lipschitz_ms", "mg_build_ms") else int(v)


def _stats_tuple(stats) -> tuple:
    return tuple(getattr(stats, f) for f in _STATS_FIELDS)


def _path_result(betas, gn, infos, K, stats) -> PathResult:
    return PathResult(betas, gn, infos, stats if isinstance(stats, tuple) else _stats_tuple(stats))


_live_engines: "weakref.WeakSet[Engine]" = weakref.WeakSet()


class Engine:
    """An engine owns a HIP stream and, optionally, an RCCL communicator.  ``get_engine`` hands out one per
    (process, device); further ones on the same device (``Engine(device_id)``) are further streams: solves on
    different engines run side by side -- the launches between the passes of one beside the passes of the other."""

    def __init__(self, device_id: int = 0):
        self._lib = load_library()
        h = C.c_void_p()
        _check(self._lib.slm_engine_create(int(device_id), C.byref(h)))
        self._h = h
        self.device_id = int(device_id)
        self._datasets = weakref.WeakSet()  # closed before the engine goes (their handles point at it)
        self._pid = os.getpid()
        _live_engines.add(self)  # (every engine, not only the per-device defaults, is released in order at exit)

    def close(self):
        if getattr(self, "_h", None):
            for ds in list(getattr(self, "_datasets", ())):
                ds.close()
            self._lib.slm_engine_destroy(self._h)
            self._h = None

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(self._lib.slm_engine_synchronize(self._h))

    def device_info(self) -> dict:
        out = (C.c_int64 * 6)()
        name = C.create_string_buffer(256)
        _check(self._lib.slm_engine_device_info(self._h, out, name, 256))
        return {
            "name": name.value.decode(),
            "compute_units": out[0],
            "lds_bytes_per_cu": out[1],
            "hbm_total_bytes": out[2],
            "hbm_free_bytes": out[3],
            "wavefront": out[4],
            "clock_khz": out[5],
        }

    # -- datasets -----------------------------------------------------------------------------
    def dataset(self, X, y, row_weight=None) -> "Dataset":
        _sync_knobs()
        X = np.asarray(X, dtype=np.float64)
        if X.ndim != 2:
            raise ValueError("X must be 2-D")
        if not (X.flags.c_contiguous or X.flags.f_contiguous):
            X = np.ascontiguousarray(X)
        n, p = X.shape
        y = _f64(y, "y", (n,))
        rw = None if row_weight is None else _f64(row_weight, "row_weight", (n,))
        rs, cs = (X.strides[0] // 8, X.strides[1] // 8)
        if X.flags.c_contiguous:
            rs, cs = p, 1
        elif X.flags.f_contiguous:
            rs, cs = 1, n
        b = load_binding()
        if b is not None:
            return Dataset(self, C.c_void_p(b.dataset_create(self._h.value, X, y, rw)), n, p)
        h = C.c_void_p()
        _check(self._lib.slm_dataset_create(self._h, _ptr(X), n, p, rs, cs, _ptr(y), _ptr(rw), C.byref(h)))
        return Dataset(self, h, n, p)

    def dataset_from_device(self, dX_ptr: int, n: int, p: int, ld: int, dy_ptr: int, drw_ptr: int = 0):
        _sync_knobs()
        h = C.c_void_p()
        _check(
            self._lib.slm_dataset_create_device(
                self._h, C.c_void_p(dX_ptr), n, p, ld, C.c_void_p(dy_ptr), C.c_void_p(drw_ptr or None), C.byref(h)
            )
        )
        return Dataset(self, h, n, p)

    def synthetic_dataset(self, n, p, seed, coef, noise_sd=0.0, row_offset=0) -> "Dataset":
        _sync_knobs()
        coef = _f64(coef, "coef", (p,))
        h = C.c_void_p()
        _check(
            self._lib.slm_dataset_create_synthetic(
                self._h, n, p, C.c_uint64(seed), row_offset, _ptr(coef), float(noise_sd), C.byref(h)
            )
        )
        return Dataset(self, h, n, p)

    # -- row-sharded mode -----------------------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        _check(self._lib.slm_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, rank: int, n_ranks: int, unique_id: bytes):
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("unique_id must be 128 bytes")
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(self._lib.slm_comm_init(self._h, rank, n_ranks, buf))

    def comm_destroy(self):
        _check(self._lib.slm_comm_destroy(self._h))

    def comm_info(self) -> tuple[int, int]:
        """(rank, ranks) of the communicator as RCCL reports them; (0, 1) without one."""
        r, n = C.c_int32(), C.c_int32()
        _check(self._lib.slm_comm_info(self._h, C.byref(r), C.byref(n)))
        return int(r.value), int(n.value)

    def comm_ranks(self) -> int:
        return self.comm_info()[1]

    def comm_collectives(self) -> int:
        """All-reduces this engine has entered since its communicator was created."""
        n = C.c_int64()
        _check(self._lib.slm_comm_collectives(self._h, C.byref(n)))
        return int(n.value)

    def comm_all_reduce_us(self, count: int, reps: int = 20) -> float:
        """Microseconds per all-reduce of ``count`` doubles on this engine's communicator (every rank calls it alike)."""
        us = C.c_double()
        _check(self._lib.slm_comm_all_reduce_probe(self._h, int(count), int(reps), C.byref(us)))
        return float(us.value)

    # -- diagnostics ------------------------------------------------------------------------------
    def dense_spd_solve(self, H, rhs):
        """(H^-1 rhs, lambda_min estimate) by the model solver's one-workgroup Cholesky (m <= 512)."""
        H = _f64(H, "H")
        m = H.shape[0]
        if H.shape != (m, m):
            raise ValueError("H must be square")
        rhs = _f64(rhs, "rhs", (m,))
        x = np.empty(m)
        mu = C.c_double()
        _check(self._lib.slm_dense_spd_solve(self._h, _ptr(H), m, _ptr(rhs), _ptr(x), C.byref(mu)))
        return x, mu.value


class Dataset:
    """Device-resident (X, y[, row weights][, groups]) plus the solver state that goes with it."""

    def __init__(self, engine: Engine, handle, n: int, p: int):
        self.engine = engine
        self._lib = engine._lib
        self._h = handle
        self.n, self.p = int(n), int(p)
        self.n_groups = self.p
        engine._datasets.add(self)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.slm_dataset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def ld(self) -> int:
        n, p, ld = C.c_int64(), C.c_int64(), C.c_int64()
        _check(self._lib.slm_dataset_shape(self._h, C.byref(n), C.byref(p), C.byref(ld)))
        return ld.value

    def download(self, want_X=True, want_y=True):
        X = np.empty((self.n, self.p)) if want_X else None
        y = np.empty(self.n) if want_y else None
        _check(self._lib.slm_dataset_download(self._h, _ptr(X), _ptr(y)))
        return X, y

    def nonfinite(self) -> int:
        """``slm_dataset_nonfinite``: 0 when the uploaded X is finite throughout, bit 0 for a NaN, bit 1 for an infinity."""
        kind = C.c_int32()
        _check(self._lib.slm_dataset_nonfinite(self._h, C.byref(kind)))
        return int(kind.value)

    def center(self):
        """Centre the device copy of (X, y) in place by the row-weighted means; returns (x_mean, y_mean)."""
        _sync_knobs()
        xm = np.empty(self.p)
        ym = C.c_double()
        _check(self._lib.slm_dataset_center(self._h, _ptr(xm), C.byref(ym)))
        return xm, ym.value

    def set_row_weights(self, row_weight):
        rw = None if row_weight is None else _f64(row_weight, "row_weight", (self.n,))
        _check(self._lib.slm_dataset_set_row_weights(self._h, _ptr(rw)))

    def clone(self, engine: "Engine | None" = None) -> "Dataset":
        """A copy of this dataset (X, y, row weights; device to device) on ``engine`` -- default: a new engine
        on the same device, i.e. a further stream.  Group structure is set again by the caller."""
        eng = Engine(self.engine.device_id) if engine is None else engine
        h = C.c_void_p()
        _check(self._lib.slm_dataset_clone(self._h, eng._h, C.byref(h)))
        return Dataset(eng, h, self.n, self.p)

    def set_targets(self, y):
        """Replace y on the device (X and what was derived from it stay)."""
        yv = _f64(y, "y", (self.n,))
        _check(self._lib.slm_dataset_set_targets(self._h, _ptr(yv)))

    def set_global_rows(self, n_global: int):
        _check(self._lib.slm_dataset_set_global_rows(self._h, int(n_global)))

    def set_groups(self, gidx, n_groups=None):
        """``gidx``: dense group index per feature (0..G-1, reference order model/_lasso.py:248)."""
        if gidx is None:
            _check(self._lib.slm_dataset_set_groups(self._h, None, 0))
            self.n_groups = self.p
            return
        g = np.ascontiguousarray(gidx, dtype=np.int32)
        if g.shape != (self.p,):
            raise ValueError(f"gidx must have shape ({self.p},)")
        G = int(g.max()) + 1 if n_groups is None else int(n_groups)
        _check(self._lib.slm_dataset_set_groups(self._h, _ptr(g), G))
        self.n_groups = G

    def max_lanes(self, flags: int = 0) -> int:
        """Lanes one ``solve_lanes`` call can take here (16 in working-set solves on large X, else what the
        fused kernel table has for this p)."""
        _sync_knobs()
        out = C.c_int32()
        _check(self._lib.slm_dataset_max_lanes(self._h, int(flags), C.byref(out)))
        return int(out.value)

    def path_lanes(self, n_points: int, flags: int = 0) -> int:
        """The lane count ``solve_path(..., lanes=0)`` runs a path of ``n_points`` on (``slm_dataset_path_lanes``)."""
        _sync_knobs()
        out = C.c_int32()
        _check(self._lib.slm_dataset_path_lanes(self._h, int(n_points), int(flags), C.byref(out)))
        return int(out.value)

    def lipschitz(self) -> float:
        _sync_knobs()
        L = C.c_double()
        _check(self._lib.slm_dataset_lipschitz(self._h, C.byref(L)))
        return L.value

    def gradient(self, z=None, reps: int = 0, split: bool = False, lanes: int = 1, lane: int = 0, probe_lanes: int = 1,
                 xtr_only: bool = False):
        """g = X^T W (X z - y)/n, loss = 1/(2n)||Xz - y||_W^2[, mean kernel ms over ``reps`` launches].

        ``split=True``: by the split pass (residuals from X, then X^T R for all lane slots on one read) with ``lanes`` lanes
        all standing at z, the gradient of lane ``lane`` returned -- how the parity tests reach xtr18 / xtr20 / xtr32 and the
        rowdot kernels through the boundary (slm_gradient_ex); ``probe_lanes`` / ``xtr_only``: the timed launches."""
        _sync_knobs()
        zz = None if z is None else _f64(z, "z", (self.p,))
        g = np.empty(self.p)
        loss = C.c_double()
        ms = C.c_double()
        if not split and lanes == 1 and probe_lanes == 1:
            _check(self._lib.slm_gradient(self._h, _ptr(zz), _ptr(g), C.byref(loss), int(reps), C.byref(ms) if reps > 0 else None))
        else:
            o = _GradientOpts(1 if split else 0, int(lanes), int(lane), int(probe_lanes), 1 if xtr_only else 0)
            _check(self._lib.slm_gradient_ex(self._h, _ptr(zz), C.byref(o), _ptr(g), C.byref(loss), int(reps),
                                             C.byref(ms) if reps > 0 else None))
        return (g, loss.value, ms.value) if reps > 0 else (g, loss.value)

    def eval_sse(self, Z, row_weight=None, sparse=None) -> np.ndarray:
        """sum_i w_i (x_i . Z[k] - y_i)^2 for every row Z[k] of ``Z`` (m, p); ``row_weight`` is e.g. the
        test mask of a CV fold.  Rows with a joint support of at most 512 columns are scored from those
        columns (``sparse=False`` forces the dense route: ceil(m/4) passes over the resident X)."""
        _sync_knobs()
        Z = _f64(np.atleast_2d(Z), "Z")
        if Z.shape[1] != self.p:
            raise ValueError(f"Z must have {self.p} columns")
        rw = None if row_weight is None else _f64(row_weight, "row_weight", (self.n,))
        out = np.empty(Z.shape[0])
        # sparse rows (the solutions of a path): gather the union of their supports once on the device
        # and score from those columns instead of passes over X
        cols = np.flatnonzero(np.any(Z != 0.0, axis=0)).astype(np.int32)
        if cols.size == 0:
            cols = np.zeros(1, dtype=np.int32)
        if cols.size <= WS_COLUMNS and sparse is not False:
            Zs = np.ascontiguousarray(Z[:, cols])
            _check(self._lib.slm_eval_sse_sparse(self._h, _ptr(cols), int(cols.size), _ptr(Zs), Z.shape[0], _ptr(rw),
                                                 _ptr(out)))
            return out
        _check(self._lib.slm_eval_sse(self._h, _ptr(Z), Z.shape[0], _ptr(rw), _ptr(out)))
        return out

    def solve_lanes(
        self,
        lanes,
        tol: float = 1e-8,
        max_iter: int = 10000,
        check_every: int = 0,
        L: float = 0.0,
        flags: int = 0,
        want_group_norms: bool = False,
        extrapolate: bool = True,
        _reweighted: bool = False,
    ) -> list:
        """Solve up to MAX_LANES independent warm-started paths on ONE pass over X per iteration.

        ``lanes``: list of dicts with ``points`` (K_l, 3) and optionally ``a``, ``b``, ``d``,
        ``beta0``, ``row_weight`` (length n, e.g. a CV-fold mask), ``n_eff`` (1/n scaling, e.g. the
        number of training rows) and ``extrap`` (K_l secant factors, see ``lane_points``; default: derived from
        the points).  Returns one ``PathResult`` per lane (shared timing fields).
        """
        _sync_knobs()
        nl = len(lanes)
        if not (1 <= nl <= MAX_CELLS):
            raise ValueError(f"between 1 and {MAX_CELLS} lanes, got {nl}")
        G = self.n_groups
        b = load_binding()
        if b is not None:  # the compiled binding marshals the lanes itself
            rounds = None
            if _reweighted:
                betas, gnb, inf, ks, st, rounds = b.solve_lanes_reweighted(
                    self._h.value, list(lanes), self.n, self.p, G, float(tol), int(max_iter), int(check_every), float(L),
                    int(flags), bool(want_group_norms), _host_pool.empty)
            else:
                betas, gnb, inf, ks, st = b.solve_lanes(self._h.value, list(lanes), self.n, self.p, G, float(tol), int(max_iter),
                                                        int(check_every), float(L), int(flags), bool(want_group_norms),
                                                        bool(extrapolate), _host_pool.empty)
            infos, out, at, p = inf.view(_INFO_DTYPE), [], 0, self.p
            for K in ks:
                out.append(PathResult(betas[at * p : (at + K) * p].reshape(K, p),
                                      gnb[at * G : (at + K) * G].reshape(K, G) if want_group_norms else None, infos[at : at + K], st))
                at += K
            return (out, [int(r) for r in rounds]) if _reweighted else out
        keep = []  # keep every buffer alive for the duration of the call
        clanes = (_Lane * nl)()
        outs = []
        # one block for the coefficients of all lanes (and one for their group norms): the engine fetches buffers
        # that follow each other in one copy
        k_all = [int(np.asarray(spec["points"]).size // 3) for spec in lanes]
        beta_block = _host_pool.empty(sum(k_all) * self.p)
        gn_block = _host_pool.empty(sum(k_all) * G) if want_group_norms else None
        infos_block = np.zeros(sum(k_all), dtype=_INFO_DTYPE)  # (one block: buffers that follow each other travel together)
        at = 0
        for l, spec in enumerate(lanes):
            pts = np.ascontiguousarray(spec["points"], dtype=np.float64).reshape(-1, 3)
            K = pts.shape[0]
            if spec.get("extrap") is not None:  # (lanes that walk pieces of several paths bring their own factors)
                gam = _f64(spec["extrap"], "extrap", (K,))
            else:
                gam = path_extrapolation(pts) if extrapolate else np.zeros(K)
            cpts = _points_block(pts, gam)

            def vec(name, size):
                v = spec.get(name)
                if v is None:
                    return None
                v = np.asarray(v)
                return _f64(v if v.shape == (size,) else np.broadcast_to(v, (size,)), name)

            a_, b_, d_ = vec("a", self.p), vec("b", G), vec("d", G)
            pen = _PenaltyStruct(_ptr(a_), _ptr(b_), _ptr(d_))
            b0 = None if spec.get("beta0") is None else _f64(spec["beta0"], "beta0", (self.p,))
            rw = None if spec.get("row_weight") is None else _f64(spec["row_weight"], "row_weight", (self.n,))
            betas = beta_block[at * self.p : (at + K) * self.p].reshape(K, self.p)
            gn = gn_block[at * G : (at + K) * G].reshape(K, G) if want_group_norms else None
            infos = infos_block[at : at + K]
            at += K
            keep.append((cpts, a_, b_, d_, pen, b0, rw))
            clanes[l].pen = C.pointer(pen)
            clanes[l].points = _as(cpts, _PathPoint)
            clanes[l].n_points = K
            clanes[l].beta0 = _ptr(b0)
            clanes[l].row_weight = _ptr(rw)
            clanes[l].n_eff = int(spec.get("n_eff") or 0)
            clanes[l].betas_out = _ptr(betas)
            clanes[l].group_norms_out = _ptr(gn)
            clanes[l].infos = _as(infos, _PointInfo)
            outs.append((betas, gn, infos, K))
        opts = _SolveOpts(float(tol), int(max_iter), int(check_every), float(L), int(flags))
        stats = _SolveStats()
        if _reweighted:
            rules, rounds = (_Reweight * nl)(), (C.c_int32 * nl)()
            for l, spec in enumerate(lanes):
                coef_scale, group_scale, numerator, eps, rtol, n_coef, n_group = spec["reweight"]
                gsc = None if group_scale is None else _f64(np.broadcast_to(np.asarray(group_scale), (int(n_group),)), "group_scale")
                keep.append(gsc)
                rules[l] = _Reweight(float(coef_scale), _ptr(gsc), float(numerator), float(eps), float(rtol), int(n_coef), int(n_group))
            _check(self._lib.slm_solve_lanes_reweighted(self._h, clanes, rules, nl, C.byref(opts), C.byref(stats), rounds))
            return [_path_result(betas, gn, infos, K, stats) for betas, gn, infos, K in outs], [int(r) for r in rounds]
        _check(self._lib.slm_solve_lanes(self._h, clanes, nl, C.byref(opts), C.byref(stats)))
        return [_path_result(betas, gn, infos, K, stats) for betas, gn, infos, K in outs]

    def solve_lanes_reweighted(self, lanes, tol: float = 1e-8, max_iter: int = 10000, flags: int = 0, want_group_norms: bool = True):
        """The re-weighting loops of Adaptive* estimators inside one launch (``slm_solve_lanes_reweighted``): every lane dict
        has ``points`` = its rounds (``max_iter`` rows, normally all ones) and ``reweight`` = ``(coef_scale, group_scale or
        None, numerator, eps, tol, n_coef, n_group)``.  Returns ``(results, rounds)``: one ``PathResult`` per lane with a row
        per round, and the number of rounds each lane ran (its last solution is row ``rounds - 1``).
        ``NotImplementedError`` when the problem is not one the on-chip solver takes, or a round did not settle there: the
        caller then loops over ``solve_lanes`` itself."""
        return self.solve_lanes(lanes, tol=tol, max_iter=max_iter, flags=int(flags) | FLAG_ON_CHIP, want_group_norms=want_group_norms,
                                extrapolate=False, _reweighted=True)

    def solve_path(
        self,
        points,
        a=None,
        b=None,
        d=None,
        beta0=None,
        tol: float = 1e-8,
        max_iter: int = 10000,
        check_every: int = 0,
        L: float = 0.0,
        flags: int = 0,
        want_group_norms: bool = False,
        extrapolate: bool = True,
        lanes: int = 1,
    ) -> PathResult:
        """Warm-started path; ``points`` is (K, 3) of (sa, sb, sd) scales applied to (a, b, d).

        ``a`` (p,), ``b`` (G,), ``d`` (G,): ``None`` means all ones.  K == 1 is the reference's
        single ``_solve``.  ``extrapolate``: when all points are multiples of one penalty direction
        (an alpha path), start point k from the secant prediction through the two previous solutions.
        ``lanes`` > 1 cuts the path into that many contiguous sub-paths that advance together, one
        pass over X serving all of them (the first point of every later sub-path starts cold).
        ``lanes=0``: the engine's choice (``slm_solve_path_lanes`` with ``n_lanes = 0``: sixteen, or eighteen / twenty
        where that saves a pass over a large X).
        """
        _sync_knobs()
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        K = pts.shape[0]
        lanes = 0 if int(lanes) == 0 and K > 1 else max(1, min(int(lanes), MAX_LANES_WIDE, K))  # (the engine takes as many as the dataset's kernels serve)
        common = dict(a=a, b=b, d=d)
        kw = dict(tol=tol, max_iter=max_iter, check_every=check_every, L=L, flags=flags,
                  want_group_norms=want_group_norms, extrapolate=extrapolate)
        if lanes == 1:
            return self.solve_lanes([dict(points=pts, beta0=beta0, **common)], **kw)[0]
        # shared path: the engine splits it into `lanes` ranges and balances them by work stealing
        G = self.n_groups
        bnd = load_binding()
        if bnd is not None:
            betas, gnb, inf, st = bnd.solve_path_lanes(self._h.value, pts, self.p, G, a, b, d, beta0, int(lanes), float(tol),
                                                       int(max_iter), int(check_every), float(L), int(flags),
                                                       bool(want_group_norms), bool(extrapolate), _host_pool.empty)
            return PathResult(betas.reshape(K, self.p), gnb.reshape(K, G) if want_group_norms else None, inf.view(_INFO_DTYPE), st)
        gam = path_extrapolation(pts) if extrapolate else np.zeros(K)
        cpts = _points_block(pts, gam)
        a_ = None if a is None else _f64(np.broadcast_to(a, (self.p,)), "a")
        b_ = None if b is None else _f64(np.broadcast_to(b, (G,)), "b")
        d_ = None if d is None else _f64(np.broadcast_to(d, (G,)), "d")
        pen = _PenaltyStruct(_ptr(a_), _ptr(b_), _ptr(d_))
        b0 = None if beta0 is None else _f64(beta0, "beta0", (self.p,))
        opts = _SolveOpts(float(tol), int(max_iter), int(check_every), float(L), int(flags))
        betas = _host_pool.empty(K * self.p).reshape(K, self.p)
        gn = _host_pool.empty(K * G).reshape(K, G) if want_group_norms else None
        infos = np.zeros(K, dtype=_INFO_DTYPE)
        stats = _SolveStats()
        _check(
            self._lib.slm_solve_path_lanes(
                self._h, C.byref(pen), _as(cpts, _PathPoint), K, lanes, C.byref(opts), _ptr(b0), _ptr(betas), _ptr(gn),
                _as(infos, _PointInfo), C.byref(stats),
            )
        )
        return _path_result(betas, gn, infos, K, stats)


    def covariance(self, row_weight=None, n_eff=0):
        """``slm_dataset_covariance``: build (or find) the Gram of the row set ``(row_weight, n_eff)`` -- what a lane
        brings as ``row_weight`` / ``n_eff`` -- so that solves with ``FLAG_COVARIANCE`` take their gradients from it
        instead of reading X.  Worth it when many solves share the row set (a fold of a large grid)."""
        _sync_knobs()
        rw = None if row_weight is None else _f64(row_weight, "row_weight", (self.n,))
        _check(self._lib.slm_dataset_covariance(self._h, _ptr(rw), int(n_eff)))

    def covariance_folds(self, row_weights, n_effs):
        """``slm_dataset_covariance_folds``: the Grams of the training sets of a K-fold split at once -- where the test rows
        partition the rows, the Gram of all rows is the sum of the test rows' Grams and is never formed from X."""
        _sync_knobs()
        rws = [_f64(w, "row_weight", (self.n,)) for w in row_weights]
        if not 1 <= len(rws) <= MAX_LANES:
            raise ValueError(f"between 1 and {MAX_LANES} row sets")
        ptrs = (C.c_void_p * len(rws))(*[w.ctypes.data for w in rws])
        ne = (C.c_int64 * len(rws))(*[int(v) for v in n_effs])
        _check(self._lib.slm_dataset_covariance_folds(self._h, ptrs, ne, len(rws)))

    def covariance_folds_begin(self, row_weights, n_effs) -> bool:
        """First half of ``covariance_folds``: queues the products of THIS rank's rows (all rows without a communicator)
        and returns; False when the masks are no K-fold partition (nothing queued: use ``covariance``)."""
        _sync_knobs()
        rws = [_f64(w, "row_weight", (self.n,)) for w in row_weights]
        if not 1 <= len(rws) <= MAX_LANES:
            raise ValueError(f"between 1 and {MAX_LANES} row sets")
        ptrs = (C.c_void_p * len(rws))(*[w.ctypes.data for w in rws])
        ne = (C.c_int64 * len(rws))(*[int(v) for v in n_effs])
        started = C.c_int32()
        _check(self._lib.slm_dataset_covariance_folds_begin(self._h, ptrs, ne, len(rws), C.byref(started)))
        return bool(started.value)

    def covariance_folds_finish(self):
        """Second half: a replica among ranks sums the parts over the ranks here (the grid mode's one collective), then the
        folds' Grams are formed and filed."""
        _check(self._lib.slm_dataset_covariance_folds_finish(self._h))

    def covariance_download(self, index: int):
        """(G, c, {yy, n_eff, fp}) of Gram ``index`` (oldest first): diagnostics and tests."""
        G = np.empty((self.p, self.p))
        c = np.empty(self.p)
        sc = np.empty(4)
        _check(self._lib.slm_dataset_covariance_download(self._h, int(index), _ptr(G), _ptr(c), _ptr(sc)))
        return G, c, {"yy": float(sc[0]), "n_eff": float(sc[1]), "fingerprint": (float(sc[2]), float(sc[3]))}

    def read_ceiling(self, reps: int = 5) -> tuple[float, float]:
        """(GB/s, ms per sweep) of a read-only stream over the device copy of X: plain 16-byte loads, summed up -- the ceiling
        the passes over X are read against on this device."""
        _sync_knobs()
        gbs, ms = C.c_double(), C.c_double()
        _check(self._lib.slm_dataset_read_ceiling(self._h, int(reps), C.byref(gbs), C.byref(ms)))
        return float(gbs.value), float(ms.value)

    def model_gram(self, download: bool = False):
        """Build the model Gram of the dataset now (``csrc/mg_kernels.hpp``: ``X^T W X / n`` from an fp16 product, what lanes
        beyond the working set's 512 columns iterate on between two passes over X; solves build it themselves when they
        need it).  ``download=True`` returns it as an (ld, ld) array (tests)."""
        _sync_knobs()
        ld = (self.p + 15) // 16 * 16
        G = np.empty((ld, ld)) if download else None
        _check(self._lib.slm_dataset_model_gram(self._h, _ptr(G) if download else None))
        return G

    def set_replicated(self, replicated: bool = True):
        """On an engine with a communicator: this dataset holds ALL rows (grid mode), not a row block -- no per-pass
        collective; the communicator only carries the folds' Grams (``covariance_folds``)."""
        _check(self._lib.slm_dataset_set_replicated(self._h, int(bool(replicated))))

    def covariance_clear(self):
        """Drop every Gram built so far (later solves run over X)."""
        _check(self._lib.slm_dataset_covariance_clear(self._h))

    def covariance_count(self) -> int:
        out = C.c_int32()
        _check(self._lib.slm_dataset_covariance_count(self._h, C.byref(out)))
        return int(out.value)

    def solve_standardized_sgl(self, a, b, beta0=None, warm=False, tol=1e-8, tol_inner=0.0, max_sweeps=0,
                               max_iter=0, want_group_norms=False):
        """``slm_solve_standardized_sgl``: the splitting for ``l1 + sum_g b_g ||X_g beta_g||_2`` with all sweeps in
        one launch.  Returns ``(beta, group_norms or None, info)``; ``NotImplementedError`` when the problem is not
        one the on-chip solver takes (the caller then runs the sweeps over ``solve_lanes``)."""
        _sync_knobs()
        G = self.n_groups
        a_ = _f64(np.broadcast_to(a, (self.p,)), "a")
        b_ = _f64(np.broadcast_to(b, (G,)), "b")
        b0 = None if beta0 is None else _f64(beta0, "beta0", (self.p,))
        opts = _SolveOpts(float(tol), int(max_iter), 0, 0.0, 0)
        beta = np.empty(self.p)
        gn = np.empty(G) if want_group_norms else None
        info = np.zeros(1, dtype=_INFO_DTYPE)
        _check(
            self._lib.slm_solve_standardized_sgl(
                self._h, _ptr(a_), _ptr(b_), C.byref(opts), float(tol_inner), int(max_sweeps), _ptr(b0), int(bool(warm)),
                _ptr(beta), _ptr(gn), _as(info, _PointInfo),
            )
        )
        return beta, gn, info[0]


class _HostPool:
    """Result buffers in page-locked host memory (``slm_host_alloc``), recycled.

    The engine copies coefficients straight into whatever host pointer it is given.  Into ordinary numpy memory the
    HIP runtime locks the pages first and remembers them; when numpy later frees such an array (a 2 MB path result,
    32 MB for sixteen lanes) the driver has to tear that registration down, which now and then stalled the NEXT
    solve's first submissions for 20 ms (bench.py's config-4 leg: 0.25 s instead of 0.16 s).  Blocks from this pool
    are never unmapped while the process lives: an array handed out keeps its block until the last view of it is
    gone (``weakref.finalize`` on the exporting buffer), then the block goes back on the free list of its size
    class (powers of two from 64 KiB).  At most ``cap`` bytes sit idle; anything beyond goes back to the driver.
    Small results, and everything after a fork, use plain numpy memory."""

    MIN_BYTES = 1 << 16

    def __init__(self, cap=1 << 30):
        self.cap, self.idle = cap, 0
        self.free: dict[int, list[int]] = {}
        # (re-entrant: `_give` runs from weakref.finalize, which a garbage collection inside `_take` / `_give` -- while
        #  the lock is held -- can trigger on this very thread)
        self.lock = threading.RLock()
        self.pid = os.getpid()

    def _take(self, size):
        with self.lock:
            blocks = self.free.get(size)
            if blocks:
                self.idle -= size
                return blocks.pop()
        ptr = C.c_void_p()
        rc = load_library().slm_host_alloc(C.c_size_t(size), C.byref(ptr))
        return ptr.value if rc == 0 and ptr.value else None

    def _give(self, ptr, size, pid):
        if pid != os.getpid() or _lib is None:  # (a forked child, or the library is already gone at exit)
            return
        with self.lock:
            if self.idle + size <= self.cap:
                self.free.setdefault(size, []).append(ptr)
                self.idle += size
                return
        try:
            _lib.slm_host_free(C.c_void_p(ptr))
        except Exception:  # interpreter shutdown
            pass

    def empty(self, n_doubles: int) -> np.ndarray:
        """Uninitialised float64 vector of `n_doubles` entries."""
        nbytes = 8 * int(n_doubles)
        if nbytes < self.MIN_BYTES or os.getpid() != self.pid or os.environ.get("SLM_NO_HOST_POOL"):
            return np.empty(int(n_doubles))
        size = 1 << (nbytes - 1).bit_length()
        ptr = self._take(size)
        if ptr is None:  # no page-locked memory to be had: ordinary memory works too
            return np.empty(int(n_doubles))
        buf = (C.c_char * size).from_address(ptr)
        weakref.finalize(buf, self._give, ptr, size, self.pid)
        return np.frombuffer(buf, dtype=np.float64, count=int(n_doubles))


_host_pool = _HostPool()


def init_local_comm(engines, timeout_s: float = 0.0) -> None:
    """Make ``engines`` (same process, same device, one host thread each) the ranks of a row-sharded job."""
    lib = load_library()
    arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
    _check(lib.slm_comm_init_local(arr, len(engines), float(timeout_s)))


# -- default engine per process / device ------------------------------------------------------------
_engines: dict[tuple[int, int], Engine] = {}
_engines_lock = threading.Lock()


def _close_engines():
    with _engines_lock:
        pid = os.getpid()
        _engines.clear()
    for eng in list(_live_engines):
        if eng._pid == pid:  # (a forked child never touches its parent's handles)
            eng.close()


import atexit  # noqa: E402

atexit.register(_close_engines)


# The library reads the SLM_* environment variables once.  The Python layer notices when one of them changes afterwards
# (os.environ[...] = ..., del os.environ[...], pytest's monkeypatch.setenv: all raise the audit events os.putenv /
# os.unsetenv BEFORE the change lands) and has the library read them again at the next call into it -- so tests and tools
# that flip a knob mid-process keep working without calling reload_knobs() themselves.
_knobs_dirty = False


def _env_audit(event, args):
    global _knobs_dirty
    if event == "os.putenv" or event == "os.unsetenv":
        key = args[0]
        if (isinstance(key, bytes) and key.startswith(b"SLM_")) or (isinstance(key, str) and key.startswith("SLM_")):
            _knobs_dirty = True


sys.addaudithook(_env_audit)


def _sync_knobs() -> None:
    global _knobs_dirty
    if _knobs_dirty:
        _knobs_dirty = False
        reload_knobs()


def reload_knobs() -> None:
    """Have the library read the SLM_* environment variables again (it reads them once, when it first needs one): for tests
    and tools that change a variable after the library was loaded."""
    _check(load_library().slm_reload_knobs())


def get_engine(device_id: int | None = None) -> Engine:
    """Process-wide engine for ``device_id`` (default: ``LOCAL_RANK`` or 0).  Fork-safe by keying on pid."""
    if device_id is None:
        device_id = int(os.environ.get("SLM_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        n = device_count()
        if n > 0:
            device_id %= n
    key = (os.getpid(), int(device_id))
    with _engines_lock:
        eng = _engines.get(key)
        if eng is None:
            eng = Engine(device_id)
            _engines[key] = eng
        return eng
