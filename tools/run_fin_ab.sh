# Dev tool (GPU box): step rate at n = 1e6 / 1e5 / 1e7 without any event record, two builds of the library interleaved
# (tools/bin/<name>/libdzo_hip.so = an earlier build, PREV=<name>; "head" = the in-tree build)
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2 3; do
 for cfg in "1000000 300" "100000 300" "10000000 100"; do
  set -- $cfg
  for which in ${PREV:-prevfin} head; do
    if [ $which = head ]; then unset DZO_LIB_PATH; else export DZO_LIB_PATH=$PWD/tools/bin/$which/libdzo_hip.so; fi
    python3 bench.py --dim $1 --steps $2 --warmup 50 --no-cpu-baseline --no-two-pass --no-kernel-events 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which n=$1', 'steps/s', d['value'], 'ms/step', d['ms_per_step'], 'evals/step', d['config']['objective_evals_per_step'], 'f_end', d['config']['f_end'])"
  done
 done
done
