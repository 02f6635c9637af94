cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2 3; do for s in 2 1; do
  DZO_TUNE_POINT_SETS=$s python3 bench.py --no-cpu-baseline --no-two-pass --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sets=$s', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'pass us', d['roofline']['avg_launch_us'], 'f_end', d['config']['f_end'])"
done; done
for st in 4 6 9; do
  DZO_TUNE_POINT_SETS=1 DZO_TUNE_POINT_STAGE_ROWS=$st python3 bench.py --no-cpu-baseline --no-two-pass --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sets=1 stage=$st', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'pass us', d['roofline']['avg_launch_us'])"
done
bash tools/sq_counters.sh regrad "lbfgs_point_pass_kernel<double, 20, false, 2>"
