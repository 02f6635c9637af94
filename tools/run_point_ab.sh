# Dev tool (GPU box): headline bench, point ring vs pair ring (DZO_TUNE_POINT_RING), interleaved.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2 3; do
  for pr in 1 0; do
    DZO_TUNE_POINT_RING=$pr python3 bench.py --no-cpu-baseline --no-two-pass --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
sp=d['roofline'].get('single_pass',{})
print('point_ring=$pr', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'pass us', d['roofline']['avg_launch_us'], 'retries', sp.get('retry_passes'), 'gram fallbacks', sp.get('fallback_gram_passes'), 'evals/step', d['config']['objective_evals_per_step'])"
  done
done
