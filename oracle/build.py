"""Build the plain-C twin of the oracle (``fista_ref.c``) into ``oracle/_build/libfista_ref.so``.

TEST INFRASTRUCTURE (see oracle/__init__.py).  ``/root/reference`` is pure Python that delegates to
cvxpy (absent and un-installable here), so there is no compilable reference source and no
``oracle/_ref`` build: the reference path is "unbuildable" in the sense of DESIGN.md's Oracle section.
"""

from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libfista_ref.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "fista_ref.c")
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(src):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", src, "-o", OUT, "-lm"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
