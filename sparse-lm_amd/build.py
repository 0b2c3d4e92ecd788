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


SAN_BINDING = os.path.join(OUT_DIR, "san", "_slm_binding.so")  # the same module under ASan + UBSan (tools/sanitize.sh)


def build_binding(force: bool = False, sanitize: bool = False) -> str:
    """The compiled Python binding of the hot calls (pybind11, g++): links against libslm_hip.so next to it.
    ``sanitize``: a second copy built with ``-fsanitize=address,undefined`` under ``_lib/san/`` -- host code only, for the
    container (``SLM_BINDING_PATH`` points the loader at it; the sanitizer runtimes have to be preloaded into python)."""
    src = os.path.join(CSRC, "binding.cpp")
    out = SAN_BINDING if sanitize else BINDING
    if not force and not _stale(out, [src, OUT]):
        return out
    import sysconfig

    import pybind11

    os.makedirs(os.path.dirname(out), exist_ok=True)
    extra = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"] if sanitize else ["-O2"]
    cmd = [
        CXX, *extra, "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
        f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}",
        src, "-o", out, f"-L{OUT_DIR}", "-lslm_hip", "-Wl,-rpath,$ORIGIN" + ("/.." if sanitize else ""),
    ]
    subprocess.run(cmd, check=True)
    return out


def _binding_or_warn(force: bool) -> None:
    """The binding is an accelerator of the Python side, not a condition of the engine: without pybind11 or the Python
    headers the ctypes route (sparselm_amd/_engine.py) serves every call.  (__graft_entry__.build() asserts it is there.)"""
    try:
        build_binding(force)
        if os.environ.get("SLM_SANITIZE"):
            build_binding(force, sanitize=True)
    except (ImportError, OSError, subprocess.CalledProcessError) as exc:
        import warnings

        for stale in (BINDING, SAN_BINDING):  # (a module of another ABI must not be found later)
            if os.path.exists(stale):
                os.remove(stale)
        warnings.warn(f"the compiled binding was not built ({exc!r}); the ctypes route serves every call", RuntimeWarning)


UNITS = ("engine.hip", "engine_solve.hip", "engine_path.hip", "engine_cov.hip", "engine_mg.hip")  # handles / memory / communicators; kernel tables + launches of a pass; the solve loop; Grams; model Gram


def build(force: bool = False, extra_flags=()) -> str:
    if not force and not _stale():
        _binding_or_warn(False)
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
    _binding_or_warn(True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
