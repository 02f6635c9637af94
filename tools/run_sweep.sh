# Dev tool (GPU box): the size sweep of DESIGN.md section 5 (headline workload at other n, m).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "1000000 20" "3000000 20" "10000000 20" "30000000 20" "10000000 10" "10000000 5"; do
  set -- $cfg
  python3 bench.py --dim $1 --history $2 --no-cpu-baseline --no-two-pass 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('n=$1 m=$2', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'single pass us', r['avg_launch_us'], 'GB/s', r['achieved'], 'evals/step', d['config']['objective_evals_per_step'])"
done
