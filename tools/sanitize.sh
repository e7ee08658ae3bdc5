#!/bin/bash
# `make sanitize` — sanitizers on the host-side code, once per round (SURVEY §5's optional aux subsystem):
#   1. AddressSanitizer + UndefinedBehaviorSanitizer builds of the CPU oracle (oracle/vpf_oracle.c) and of the C / C++ harnesses around the
#      launch planners (tests/c/lzm_plan_capi.cpp, plan_bounds_capi.c, persist_capi.cpp), run under the existing CPU tests;
#   2. a ThreadSanitizer build of tests/c/tsan_harness.cpp: eight threads cycling shapes through a four-entry table arena (LzmTableCache) and
#      asking for counter slots (PersistSlotTable).
# Writes the log to profiles/r06_sanitize.txt (or $1).  No GPU.
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r06_sanitize.txt}
ASAN=$(gcc -print-file-name=libasan.so)
{
  echo "# tools/sanitize.sh on $(date -u +%Y-%m-%dT%H:%MZ), $(gcc --version | head -1)"
  echo "== 1. ASAN + UBSAN: oracle + planner harnesses under the CPU tests"
  VPF_TEST_BUILD_TAG=san VPF_TEST_CFLAGS="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g" \
  LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    timeout 3000 python -m pytest -q -x -p no:cacheprovider tests/test_lzm_plan_cpu.py tests/test_plan_bounds_cpu.py tests/test_persist_cpu.py tests/test_oracle_assumptions.py \
      tests/test_oracle_third_party.py tests/test_lanczos_error_bound.py tests/test_reference_fixtures.py "tests/test_oracle_kat.py" -m "not gpu" -k "not exhaustive" 2>&1 | tail -15
  echo "== 2. TSAN: table cache + counter slots from eight threads"
  g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -Wall -Werror -Ivideoprocessingframework_amd/csrc tests/c/tsan_harness.cpp -o tests/_build/tsan_harness && \
    TSAN_OPTIONS=halt_on_error=1 ./tests/_build/tsan_harness; echo "tsan harness exit status $?"
} 2>&1 | tee "$LOG"
