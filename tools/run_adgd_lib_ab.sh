cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2 3; do for which in prevdec head; do
if [ $which = head ]; then unset DZO_LIB_PATH; else export DZO_LIB_PATH=$PWD/tools/bin/$which/libdzo_hip.so; fi
python3 bench.py --workload adgd --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('adgd $which', d['value'], d['ms_per_step'], d['config']['f_end'], d['config']['steps_after_a_rejected_trial'])"
done; done
