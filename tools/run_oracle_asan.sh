#!/bin/bash
# tools/run_oracle_asan.sh [out file]: the CPU oracle (both precisions) built with -fsanitize=address,undefined (make -C oracle asan ->
# oracle/_asan/) and driven through its CPU tests — known-answer tests, operator build, host logic, the fp32 budget test.  CPU build only
# (GPU AddressSanitizer is not available on this pool).  The sanitizer runtime is preloaded because the host program is python.
#   bash tools/run_oracle_asan.sh profiles/r04/oracle_asan_ubsan.txt
set -u
root=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-/dev/stdout}
make -s -C "$root/oracle" asan || exit 1
asan=$(gcc -print-file-name=libasan.so)
ubsan=$(gcc -print-file-name=libubsan.so)
{
  echo "# $(gcc --version | head -1); -fsanitize=address,undefined -O1 -g; LD_PRELOAD=$asan:$ubsan"
  echo "# ASAN_OPTIONS=detect_leaks=0 (python itself leaks at exit), halt_on_error=1; UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1"
  cd "$root" && FDTD_ORACLE_DIR="$root/oracle/_asan" LD_PRELOAD="$asan:$ubsan" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
    UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 OMP_NUM_THREADS=4 \
    python -m pytest tests/test_oracle_kat_cpu.py tests/test_operator_build_cpu.py tests/test_host_logic_cpu.py tests/test_fp32_budget_cpu.py -q -x -p no:cacheprovider 2>&1
  echo "# exit code: $?"
} > "$out"
tail -3 "$out"
