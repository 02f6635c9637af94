cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2; do for b in 4 6 8; do
DZO_TUNE_ADGD_BPC=$b python3 bench.py --workload adgd --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bpc=$b', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['config']['f_end'], d['config']['steps_after_a_rejected_trial'])"
done; done
