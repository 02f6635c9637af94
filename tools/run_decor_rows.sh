# Dev tool (GPU box): the headline workload with the decorators of legacy/DZOptimization.jl:219-296 riding on the point pass
set -e
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "10000000 20 " "10000000 20 l2=0.001" "10000000 20 box=-1.15:0.95" "10000000 20 l2=0.001,box=-1.15:0.95" "10000001 20 l2=0.001,box=-1.15:0.95" "10000000 10 l2=0.001,box=-1.15:0.95" "10000000 24 l2=0.001,box=-1.15:0.95" "1000000 20 l2=0.001,box=-1.15:0.95"; do
  set -- $cfg
  for sp in 1 0; do
  DZO_TUNE_SINGLE_PASS=$sp python3 bench.py --dim $1 --history $2 --decorators "$3" --no-cpu-baseline --no-two-pass 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline'] or {'hip_event_name': None, 'avg_launch_us': None, 'achieved': None, 'frac': None}
print('n=$1 m=$2 decor=[$3] single_pass=$sp', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], r['hip_event_name'], 'us', r['avg_launch_us'], 'frac', r['frac'], 'evals/step', d['config']['objective_evals_per_step'], d['config']['history_layout'], 'stuck' if d['config']['any_stuck'] else '')"
  done
done
