"""The C-ABI library loads and exports exactly what include/slm_engine.h declares (no GPU needed)."""

import ctypes
import os
import re

import pytest

from sparselm_amd import _engine

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "slm_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slm_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert _declared_functions() == sorted(_engine.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _engine.load_library()
    for name in _declared_functions():
        assert hasattr(lib, name), name
    assert lib.slm_abi_version() == 3


def test_struct_layouts_match_header():
    # sizes the C side uses (checked against natural alignment of the header's structs)
    assert ctypes.sizeof(_engine._PathPoint) == 32
    assert ctypes.sizeof(_engine._SolveOpts) == 32
    assert ctypes.sizeof(_engine._PointInfo) == 48
    assert ctypes.sizeof(_engine._SolveStats) == 80
    assert ctypes.sizeof(_engine._PenaltyStruct) == 24


def test_no_cpu_fallback_without_device():
    if _engine.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_engine.EngineError):
        _engine.Engine(0)
    # and the estimator surface fails loudly too
    from sparselm_amd.model import Lasso

    with pytest.raises(_engine.EngineError):
        Lasso(alpha=0.1).fit([[-1.0], [0.0], [1.0]], [-1.0, 0.0, 1.0])


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_engine, "_lib", None)
    monkeypatch.setenv("SLM_HIP_LIBRARY", str(tmp_path / "nope.so"))
    with pytest.raises(_engine.EngineError, match="no CPU fallback"):
        _engine.load_library()
    monkeypatch.delenv("SLM_HIP_LIBRARY")
    monkeypatch.setattr(_engine, "_lib", None)
    _engine.load_library()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sparse-lm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "fista_ref" not in src, f


def test_path_extrapolation_factors():
    import numpy as np

    a = np.geomspace(1.0, 1e-3, 6)
    g = _engine.path_extrapolation(np.c_[a, 0 * a, 0 * a])
    assert g[0] == 0 and g[1] == 0
    np.testing.assert_allclose(g[2:], (a[2:] - a[1:-1]) / (a[1:-1] - a[:-2]))
    # sparse-group path: both scales move together -> still one direction
    g2 = _engine.path_extrapolation(np.c_[0.3 * a, 0.7 * a, 0 * a])
    np.testing.assert_allclose(g2, g)
    # penalty changes shape along the path -> no extrapolation
    assert not _engine.path_extrapolation([(0, 1, 0), (0.3, 0.7, 0), (0, 1, 1)]).any()
    # repeated point -> zero denominator handled
    assert np.all(np.isfinite(_engine.path_extrapolation([(1, 0, 0), (1, 0, 0), (0.5, 0, 0)])))
