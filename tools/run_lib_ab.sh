# Dev tool (GPU box): headline bench with several builds of libdzo_hip.so (DZO_LIB_PATH), interleaved.
#   LIBS="r2 dpp tree" ROUNDS=3 bash tools/run_lib_ab.sh     ("head" = the in-tree build; others = tools/bin/<name>/libdzo_hip.so)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
for r in $(seq 1 ${ROUNDS:-3}); do
  for which in ${LIBS:-head r2}; do
    if [ $which = head ]; then unset DZO_LIB_PATH; else export DZO_LIB_PATH=$PWD/tools/bin/$which/libdzo_hip.so; fi
    python3 bench.py --no-cpu-baseline --no-two-pass --steps ${STEPS:-100} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
sp=d['roofline'].get('single_pass',{})
print('$which', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'pass us', d['roofline']['avg_launch_us'], 'retries', sp.get('retry_passes'), 'f_end', d['config']['f_end'])"
  done
done
