set -x
cd "${GRAFT_REPO_ROOT:-.}"
DZO_LIB_PATH=$PWD/tools/bin/regrad/libdzo_hip.so python -m pytest tests/test_gpu_lbfgs.py tests/test_gpu_lbfgs_scale.py tests/test_gpu_fuzz.py -m gpu -x -q -k "point or single_pass or scale or fuzz_point or full_size or stuck or adopts" > gpurun_out/r03_t8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t8.log; tail -15 gpurun_out/r03_t8.log
LIBS="regrad head r2" ROUNDS=3 bash tools/run_lib_ab.sh
