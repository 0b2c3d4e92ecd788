"""The C-ABI library loads and exports exactly what include/slm_engine.h declares (no GPU needed)."""

import ctypes
import os
import re
import subprocess

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
    header = open(os.path.join(ROOT, "include", "slm_engine.h")).read()
    version = int(re.search(r"#define\s+SLM_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.slm_abi_version() == version == _engine.ABI_VERSION


# every struct of the header with the ctypes class that mirrors it
_STRUCTS = {
    "slm_penalty": "_PenaltyStruct",
    "slm_path_point": "_PathPoint",
    "slm_solve_opts": "_SolveOpts",
    "slm_point_info": "_PointInfo",
    "slm_solve_stats": "_SolveStats",
    "slm_lane": "_Lane",
}


def _header_layout(tmp_path):
    """sizeof / offsetof of every field of every struct, as the C compiler lays the header out."""
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "slm_engine.h")).read(), flags=re.S)
    fields = {}
    for name in _STRUCTS:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, flags=re.S).group(1)
        fields[name] = [re.search(r"(\w+)\s*$", part.strip()).group(1)
                        for decl in body.split(";") if decl.strip() for part in decl.split(",")]
    lines = ["#include <stddef.h>", "#include <stdio.h>", '#include "slm_engine.h"', "int main(void) {"]
    for name, flds in fields.items():
        lines.append(f'  printf("{name} size %zu\\n", sizeof({name}));')
        for f in flds:
            lines.append(f'  printf("{name} {f} %zu\\n", offsetof({name}, {f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    layout = {}
    for line in out.splitlines():
        name, field, value = line.split()
        layout.setdefault(name, {})[field] = int(value)
    return layout, fields


def test_struct_layouts_match_header(tmp_path):
    """Field order, offsets and sizes of the ctypes mirrors against the header itself (compiled with gcc)."""
    layout, fields = _header_layout(tmp_path)
    for name, cls_name in _STRUCTS.items():
        cls = getattr(_engine, cls_name)
        assert [f[0] for f in cls._fields_] == fields[name], name
        assert ctypes.sizeof(cls) == layout[name]["size"], name
        for f in fields[name]:
            assert getattr(cls, f).offset == layout[name][f], (name, f)
    # the numpy view of slm_point_info used on the way out
    assert _engine._INFO_DTYPE.itemsize == layout["slm_point_info"]["size"]
    for f in fields["slm_point_info"]:
        assert _engine._INFO_DTYPE.fields[f][1] == layout["slm_point_info"][f], f


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
