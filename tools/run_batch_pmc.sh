# Dev tool (GPU box): FETCH_SIZE / WRITE_SIZE of the batched BFGS step kernel (separate rocprofv3 --pmc passes).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/batch_pmc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/batch_pmc_$C -o f -- python3 bench.py --workload bfgs_batched --steps 20 --no-cpu-baseline > gpurun_out/batch_pmc_$C.log 2>&1 || true
  python3 - "$C" <<'PY'
import csv, glob, sys
c = sys.argv[1]
f = glob.glob(f'gpurun_out/batch_pmc_{c}/**/*counter_collection.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if 'batch_step' in r['Kernel_Name'] and r['Counter_Name'] == c]
print(c, 'KiB per dispatch (10 steps x 1024 instances each):', [round(float(r['Counter_Value'])) for r in rows])
PY
done
