set -x
cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests -m gpu -q -x -k "one_register_set or fused_reduce_finish" > gpurun_out/r03_t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t7.log; tail -6 gpurun_out/r03_t7.log
for r in 1 2; do for lz in 1 0; do
  DZO_TUNE_LAZY_D=$lz python3 bench.py --no-cpu-baseline --no-two-pass --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('lazy_d=$lz', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'pass us', k['lbfgs_single_pass']['avg_us'], 'retry us', k.get('lbfgs_single_pass_retry',{}).get('avg_us'), 'retries', k.get('lbfgs_single_pass_retry',{}).get('launches'))"
done; done
bash tools/collect_profiles.sh r03 > gpurun_out/r03_collect.log 2>&1; tail -15 gpurun_out/r03_collect.log
