"""Build the HIP engine (gfx950 only) into ``sparselm_amd/_lib/libslm_hip.so``.

``hipcc`` cross-compiles without a GPU, so this runs in CI / a GPU-less container too.
Usage: ``python sparse-lm_amd/build.py [--force]``.
"""

from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "sparselm_amd", "_lib")
OUT = os.path.join(OUT_DIR, "libslm_hip.so")
BINDING = os.path.join(OUT_DIR, "_slm_binding.so")  # pybind11 module over the C ABI (csrc/binding.cpp): host code only
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", "g++")


def _stale(out=OUT, deps=None) -> bool:
    if not os.path.exists(out):
        return True
    if deps is None:
        deps = [d for d in glob.glob(os.path.join(CSRC, "*")) if not d.endswith("binding.cpp")]
    deps = list(deps) + [os.path.join(HERE, "..", "include", "slm_engine.h")]
    newest = max(os.path.getmtime(d) for d in deps)
    return newest > os.path.getmtime(out)


def build_binding(force: bool = False) -> str:
    """The compiled Python binding of the hot calls (pybind11, g++): links against libslm_hip.so next to it."""
    src = os.path.join(CSRC, "binding.cpp")
    if not force and not _stale(BINDING, [src, OUT]):
        return BINDING
    import sysconfig

    import pybind11

    cmd = [
        CXX, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
        f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}",
        src, "-o", BINDING, f"-L{OUT_DIR}", "-lslm_hip", "-Wl,-rpath,$ORIGIN",
    ]
    subprocess.run(cmd, check=True)
    return BINDING


UNITS = ("engine.hip", "engine_solve.hip", "engine_cov.hip", "engine_mg.hip")  # handles / memory / communicators; solve loop; Grams; model Gram


def build(force: bool = False, extra_flags=()) -> str:
    if not force and not _stale():
        build_binding(False)
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    obj_dir = os.path.join(HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    from concurrent.futures import ThreadPoolExecutor

    def compile_unit(name):
        obj = os.path.join(obj_dir, name.replace(".hip", ".o"))
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra_flags, "-c", os.path.join(CSRC, name), "-o", obj],
                       check=True)
        return obj

    with ThreadPoolExecutor(max_workers=len(UNITS)) as pool:  # (the units compile side by side: 38 s instead of 60)
        objs = list(pool.map(compile_unit, UNITS))
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT, "-ldl"], check=True)
    build_binding(True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
