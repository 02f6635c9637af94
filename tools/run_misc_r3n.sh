cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests -m gpu -q -x -k "adgd or AdGD" > gpurun_out/r03_t9.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t9.log; tail -6 gpurun_out/r03_t9.log
for r in 1 2; do
python3 bench.py --workload adgd --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('adgd', d['value'], 'ms/step', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()}, d['config']['f_end'])"
done
