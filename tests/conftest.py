"""Shared test setup: import paths, the ``gpu`` marker, reference-style fixtures.

Fixtures mirror /root/reference/tests/conftest.py (seeded rng, 25-sample random models with 20 and
30 features, group labels aligned with the active features) so that the surface tests read like the
reference's own.
"""

import os
import sys

# The CPU oracle leans on multi-threaded BLAS; on a many-core host whose cores are shared or
# throttled, 256 BLAS threads turn a two-second oracle fit into minutes.  (Set before numpy loads.)
for _var in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_var, "16")

import numpy as np
import pytest
from sklearn.datasets import make_regression

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for path in (ROOT, os.path.join(ROOT, "sparse-lm_amd")):
    if path not in sys.path:
        sys.path.insert(0, path)

SEED = 0
N_FEATURES = [20, 30]  # an overdetermined and an underdetermined case
N_SAMPLES = 25
N_INFORMATIVE = 10


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    _ensure_library()


def _ensure_library():
    """The product never builds or falls back by itself (a missing library is an EngineError); the test
    session compiles it once if the tree was shipped without the built file (hipcc cross-compiles anywhere)."""
    try:
        from sparselm_amd import _engine

        if os.environ.get("SLM_HIP_LIBRARY") or os.path.exists(_engine.library_path()):
            return
        import importlib.util

        spec = importlib.util.spec_from_file_location("_slm_engine_build", os.path.join(ROOT, "sparse-lm_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(force=False)
    except Exception as exc:  # the ABI tests then report the missing library themselves
        print(f"[conftest] could not build libslm_hip.so: {exc}", file=sys.stderr)


def _gpu_present() -> bool:
    try:
        from sparselm_amd import _engine

        return _engine.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "lasso_family_golden.npz")
    with np.load(path) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="package")
def rng():
    return np.random.default_rng(SEED)


@pytest.fixture(scope="package", params=N_FEATURES)
def random_model(rng, request):
    X, y, beta = make_regression(
        n_samples=N_SAMPLES,
        n_features=request.param,
        n_informative=N_INFORMATIVE,
        coef=True,
        random_state=int(rng.integers(0, 2**32 - 1)),
        bias=10 * rng.random(),
    )
    return X, y, beta


@pytest.fixture(params=[4, 6], scope="package")
def random_model_with_groups(random_model, rng, request):
    """Groups consistent with the active features (same construction idea as the reference fixture)."""
    X, y, beta = random_model
    n_groups = request.param
    n_active_groups = n_groups // 3 + 1
    per_group = len(beta) // n_groups
    active_groups = rng.choice(range(n_groups), size=n_active_groups, replace=False)
    inactive_groups = np.setdiff1d(range(n_groups), active_groups)
    groups = np.zeros(len(beta), dtype=int)
    active = np.where(abs(beta) > 0)[0]
    inactive = np.setdiff1d(np.arange(len(beta)), active)
    for pool_name, gids in (("active", active_groups), ("inactive", inactive_groups)):
        for i in gids:
            pool = active if pool_name == "active" else inactive
            inds = rng.choice(pool, size=per_group, replace=False) if len(pool) > per_group else pool
            groups[inds] = i
            if pool_name == "active":
                active = np.setdiff1d(active, inds)
            else:
                inactive = np.setdiff1d(inactive, inds)
    return X, y, beta, groups
