set -x
cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests -m gpu -q -x > gpurun_out/r03_full2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_full2.log; tail -4 gpurun_out/r03_full2.log
for w in bfgs_dense lbfgs_lse_f32 adgd bfgs_batched; do
  python3 bench.py --workload $w --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w', d['value'], d['unit'], 'ms/step', d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done
LIBS="head r2" ROUNDS=2 bash tools/run_lib_ab.sh
