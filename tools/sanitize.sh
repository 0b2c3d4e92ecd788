#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the host code that runs without a GPU (GPU sanitizers are
# not available on the pool, and this script is for the BUILD CONTAINER only -- never the GPU box):
#   1. csrc/host_logic.hpp -- the engine's device-free bookkeeping, the file engine*.hip include -- through tests/host_logic_test.cpp;
#   2. csrc/binding.cpp -- the pybind11 module of the hot calls -- built with the sanitizers (sparse-lm_amd/build.py, SLM_SANITIZE=1)
#      and driven by tests/test_binding_cpu.py and tests/test_abi.py with the runtimes preloaded into python.
# usage: bash tools/sanitize.sh [log]      (default log: profiles/r06_sanitizers.log)
set -o pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$R/profiles/r06_sanitizers.log}
CXX=${CXX:-g++}
TMP=$(mktemp -d)
{
  echo "== sanitizers (container, $(date -u +%Y-%m-%dT%H:%MZ)); $($CXX --version | head -1)"
  echo "== 1. host_logic_test under -fsanitize=address,undefined (leak check on)"
  $CXX -std=c++17 -O1 -g -Wall -Wextra -Werror -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer \
      "$R/tests/host_logic_test.cpp" -o "$TMP/host_logic_test" && ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 "$TMP/host_logic_test"
  echo "exit code: $?"
  echo "== 2. binding.cpp under the sanitizers: tests/test_binding_cpu.py tests/test_abi.py"
  SLM_SANITIZE=1 python "$R/sparse-lm_amd/build.py" > /dev/null && ls -la "$R/sparse-lm_amd/sparselm_amd/_lib/san/_slm_binding.so"
  # (python itself is not built for the leak checker: leaks off; every other report of either sanitizer ends the run)
  LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
  ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  SLM_BINDING_PATH="$R/sparse-lm_amd/sparselm_amd/_lib/san/_slm_binding.so" SLM_EXPECT_SANITIZED_BINDING=1 \
  python -m pytest "$R/tests/test_binding_cpu.py" "$R/tests/test_abi.py" -q -p no:cacheprovider 2>&1 | tail -15
  echo "exit code: ${PIPESTATUS[0]}"
} 2>&1 | tee "$LOG"
rm -rf "$TMP" "$R/sparse-lm_amd/sparselm_amd/_lib/san"   # (the sanitized module is for this script only: it does not travel to the GPU box)
