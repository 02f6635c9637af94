# Dev tool (GPU box): headline bench, this build vs tools/bin/old/libdzo_hip.so (DZO_LIB_PATH), interleaved.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then export DZO_LIB_PATH=$PWD/tools/bin/old/libdzo_hip.so; else unset DZO_LIB_PATH; fi
    python3 bench.py --no-cpu-baseline --no-two-pass --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'pass us', d['roofline']['avg_launch_us'])"
  done
done
