# A/B of the AdGD decision: in the next pass's prologue (default) against a kernel of its own (DZO_TUNE_ADGD_PROLOGUE=0)
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2 3; do for which in 1 0; do
export DZO_TUNE_ADGD_PROLOGUE=$which
python3 bench.py --workload adgd --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('adgd prologue=$which', d['value'], d['ms_per_step'], d['config']['f_end'], d['config']['steps_after_a_rejected_trial'])"
done; done
