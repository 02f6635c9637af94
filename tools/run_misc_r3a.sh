set -x
cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r03_full1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_full1.log; tail -14 gpurun_out/r03_full1.log
LIBS="head tree_refill r2" ROUNDS=2 bash tools/run_lib_ab.sh
for s in 2 1 2 1; do
  DZO_TUNE_POINT_SETS=$s python3 bench.py --no-cpu-baseline --no-two-pass --steps 100 --history 16 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('m=16 sets=$s', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'pass us', d['roofline']['avg_launch_us'])"
done
BENCH_DIST_BACKEND=gloo BENCH_FORCE_DEVICE=0 python3 bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/r03_gloo2.json 2> gpurun_out/r03_gloo2.err; echo "gloo2 rc=$?"; tail -c 3000 gpurun_out/r03_gloo2.json; tail -5 gpurun_out/r03_gloo2.err
