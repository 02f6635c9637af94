# A/B of the dense quadratic's search rounds: DZO_TUNE_PHI6_COLS = 1 (one column per block) / 2 / 4 (columns per block, two sets ahead)
# / 22 (two columns, two sets ahead, three blocks per CU)
cd "${GRAFT_REPO_ROOT:-.}"
for r in 1 2; do for which in ${WHICH:-1 2 22 4}; do
export DZO_TUNE_PHI6_COLS=$which
python3 bench.py --workload bfgs_dense --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bfgs_dense cols=$which', d['value'], d['ms_per_step'], 'phi us', d['kernels']['objective_quadratic_phi']['avg_us'])"
done; done
