cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests -m gpu -q -x > gpurun_out/r03_full3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_full3.log; tail -8 gpurun_out/r03_full3.log
LIBS="head r2" ROUNDS=2 bash tools/run_lib_ab.sh
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; tail -c 1500 gpurun_out/r03_bench_default.json
