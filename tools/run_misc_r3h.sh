cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests -m gpu -q -x > gpurun_out/r03_full4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_full4.log; tail -12 gpurun_out/r03_full4.log
LIBS="head r2" ROUNDS=2 bash tools/run_lib_ab.sh
for st in 8 12 16 18; do
  DZO_TUNE_POINT_STAGE_ROWS=$st python3 bench.py --no-cpu-baseline --no-two-pass --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stage=$st', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'pass us', d['roofline']['avg_launch_us'])"
done
