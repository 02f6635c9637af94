cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests -m gpu -q -x > gpurun_out/r03_full5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_full5.log; tail -12 gpurun_out/r03_full5.log
for r in 1 2 3; do
  python3 bench.py --no-cpu-baseline --no-two-pass --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('head', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'pass us', d['roofline']['avg_launch_us'], 'f_end', d['config']['f_end'])"
done
