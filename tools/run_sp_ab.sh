# Dev tool (GPU box): interleaved A/B of single-pass kernel variants (DZO_TUNE_SP_DEBUG masks) with
# HIP-event times and, from separate rocprofv3 --pmc passes, FETCH_SIZE / WRITE_SIZE per dispatch.
#   bash tools/run_sp_ab.sh "0,256"
set -e
cd "${GRAFT_REPO_ROOT:-.}"
MASKS="${1:-0,256}"
export TMPDIR=/tmp
AB_MASKS=$MASKS AB_ROUNDS=6 python3 tools/sp_ablate.py 2>/dev/null | tail -4
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/sp_pmc_$C
  AB_MASKS=$MASKS AB_ROUNDS=3 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/sp_pmc_$C -o f -- python3 tools/sp_ablate.py > gpurun_out/sp_pmc_$C.log 2>&1 || true
  python3 - "$C" <<'PY'
import csv, glob, sys
c = sys.argv[1]
f = glob.glob(f'gpurun_out/sp_pmc_{c}/**/*counter_collection.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if 'single_pass' in r['Kernel_Name'] and r['Counter_Name'] == c]
print(c, 'KiB per dispatch, in launch order:', [round(float(r['Counter_Value'])) for r in rows])
PY
done
