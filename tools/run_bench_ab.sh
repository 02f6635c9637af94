# Dev tool (GPU box): the headline bench under several DZO_TUNE_SP_DEBUG masks, interleaved, short lines.
#   bash tools/run_bench_ab.sh "0 1024"
set -e
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2; do
  for m in ${1:-0 1024}; do
    DZO_TUNE_SP_DEBUG=$m python3 bench.py --no-cpu-baseline --no-two-pass --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mask $m', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'single pass us', d['roofline']['avg_launch_us'], 'evals/step', d['config']['objective_evals_per_step'])"
  done
done
