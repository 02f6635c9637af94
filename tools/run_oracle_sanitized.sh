#!/bin/bash
# CPU only: the oracle's C restatement under AddressSanitizer + UndefinedBehaviorSanitizer, driven by the oracle's own
# tests and the golden vectors (GPU sanitizers are not available on the pool; the host side that CAN be checked is this).
set -e
cd "$(dirname "$0")/.."
make -C oracle -s sanitize
export DZO_ORACLE_LIB=$PWD/oracle/libdzo_oracle_san.so
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-4}
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    python3 -m pytest tests/test_oracle.py tests/test_oracle_safeguards.py tests/test_golden.py tests/test_decorators.py -q -m "not gpu" -p no:cacheprovider "$@"
