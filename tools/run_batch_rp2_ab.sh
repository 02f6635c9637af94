# Dev tool (GPU box): the batched kernel's RP = 2 instantiation (256 < n <= 512) in three builds, interleaved
set -e
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2; do
  for which in head rp2_bpc1 rp2_uj2; do
    if [ $which = head ]; then unset DZO_LIB_PATH; else export DZO_LIB_PATH=$PWD/tools/bin/$which/libdzo_hip.so; fi
    for n in 512 384; do
    python3 bench.py --workload bfgs_batched --dim $n --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which n=$n', 'inst-steps/s', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], d['kernels'])"
    done
  done
done
