"""csrc/host_logic.hpp -- the engine's device-free bookkeeping (pool ledger, row sets, interleaved lane walks, launch
geometry, Gram look-up, triangle tiles) -- compiled with g++ and run on the CPU: tests/host_logic_test.cpp.  The CPU suite
runs it plainly; tools/sanitize.sh runs the same binary under AddressSanitizer + UndefinedBehaviorSanitizer (and the compiled
binding with the binding tests) and keeps the log under profiles/."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("sanitize", [False, True])
def test_host_logic(tmp_path, sanitize):
    cxx = os.environ.get("CXX", "g++")
    if shutil.which(cxx) is None:
        pytest.skip("no C++ compiler")
    exe = tmp_path / "host_logic_test"
    flags = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"] if sanitize else []
    build = subprocess.run([cxx, "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", *flags, os.path.join(ROOT, "tests", "host_logic_test.cpp"),
                            "-o", str(exe)], capture_output=True, text=True)
    if sanitize and build.returncode != 0 and "sanitize" in build.stderr.lower():
        pytest.skip("this toolchain has no sanitizer runtime")
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "host_logic_test: ok" in run.stdout
