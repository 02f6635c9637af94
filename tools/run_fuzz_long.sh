# GPU box: the fuzz scripts with many cases and fresh seeds (dev tool)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
set -e -o pipefail
FUZZ_CASES=${CASES:-250} FUZZ_SEED=${SEED:-11} timeout -k 10 500 python3 tests/fuzz_bfgs_search.py 2>&1 | tail -3
FUZZ_CASES=${CASES:-250} FUZZ_SEED=${SEED:-12} timeout -k 10 500 python3 tests/fuzz_points.py 2>&1 | tail -3
FUZZ_CASES=${CASES:-250} FUZZ_SEED=${SEED:-13} timeout -k 10 500 python3 tests/fuzz_lbfgs.py 2>&1 | tail -3
FUZZ_CASES=${CASES:-250} FUZZ_SEED=${SEED:-14} timeout -k 10 500 python3 tests/fuzz_adgd.py 2>&1 | tail -3
FUZZ_CASES=$(( ${CASES:-250} / 4 )) FUZZ_SEED=${SEED:-15} timeout -k 10 700 python3 tests/fuzz_batched.py 2>&1 | tail -3
