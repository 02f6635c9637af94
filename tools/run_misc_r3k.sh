cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests -m gpu -q -x > gpurun_out/r03_full6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_full6.log; tail -12 gpurun_out/r03_full6.log
for r in 1 2; do for sp in 1 0; do
  DZO_TUNE_POINT_SPLIT=$sp python3 bench.py --no-cpu-baseline --no-two-pass --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('split=$sp', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'pass us', d['roofline']['avg_launch_us'], 'edges us', k.get('lbfgs_single_pass_edges',{}).get('avg_us'), 'f_end', d['config']['f_end'])"
done; done
python3 -c "import __graft_entry__ as g; g.smoke()"
