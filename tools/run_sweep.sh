# Dev tool (GPU box): the size sweep of DESIGN.md section 5 (headline workload at other n, m).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in "100000 20" "400000 20" "1000000 20" "3000000 20" "10000000 20" "30000000 20" "10000000 24" "10000000 21" "10000000 18" "10000000 16" "10000000 14" "10000000 12" "10000000 10" "10000000 8" "10000000 6" "10000000 5" "10000000 26" "10000001 20" "9999999 10"; do   # (m > 24 takes the two-pass kernels; the ragged sizes run on the phantom-padded point ring since round 4)
  set -- $cfg
  python3 bench.py --dim $1 --history $2 --no-cpu-baseline --no-two-pass 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline'] or {'hip_event_name': None, 'avg_launch_us': None, 'achieved': None, 'frac': None}   # (None: the run got stuck before the timed region)
print('n=$1 m=$2', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], r['hip_event_name'], 'us', r['avg_launch_us'], 'GB/s', r['achieved'], 'frac', r['frac'], 'evals/step', d['config']['objective_evals_per_step'], d['config']['history_layout'], 'stuck' if d['config']['any_stuck'] else '')"
done
