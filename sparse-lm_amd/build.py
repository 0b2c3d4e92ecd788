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
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    deps = glob.glob(os.path.join(CSRC, "*")) + [os.path.join(HERE, "..", "include", "slm_engine.h")]
    newest = max(os.path.getmtime(d) for d in deps)
    return newest > os.path.getmtime(OUT)


def build(force: bool = False, extra_flags=()) -> str:
    if not force and not _stale():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = [
        HIPCC,
        "--offload-arch=gfx950",
        "-O3",
        "-std=c++17",
        "-fPIC",
        "-shared",
        *extra_flags,
        os.path.join(CSRC, "engine.hip"),
        "-o",
        OUT,
        "-ldl",
    ]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
