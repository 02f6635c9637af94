cd "${GRAFT_REPO_ROOT:-.}"
run() { python3 bench.py --no-cpu-baseline --no-two-pass --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'pass us', d['roofline']['avg_launch_us'])"; }
run default
DZO_TUNE_POINT_SETS=2 run sets2
DZO_TUNE_POINT_STAGE_ROWS=18 run stage18
DZO_TUNE_POINT_STAGE_ROWS=8 run stage8
DZO_TUNE_POINT_PLAIN_MB=200 run plainstores
DZO_TUNE_STREAM_MAJOR=0 run tilemajor
run default
bash tools/collect_profiles.sh r03 > gpurun_out/r03_collect4.log 2>&1; tail -3 gpurun_out/r03_collect4.log
