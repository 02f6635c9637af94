set -x
cd "${GRAFT_REPO_ROOT:-.}"
python -m pytest tests -m gpu -q -x -k "config4 or two_loop or each_step or lse or split or callback or deterministic" > gpurun_out/r03_t6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t6.log; tail -4 gpurun_out/r03_t6.log
for ff in 1 0 1 0; do
  DZO_TUNE_FUSED_FINISH=$ff python3 bench.py --workload lbfgs_lse_f32 --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fused_finish=$ff', 'dirs/s', d['value'], 'wall us', d['roofline']['wall_us_per_direction'], 'kernel sum', d['roofline']['kernel_sum_us'], {k:v['avg_us'] for k,v in d['kernels'].items()})"
done
export TMPDIR=/tmp
rm -rf gpurun_out/lse_trace; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lse_trace -o t -- python3 bench.py --workload lbfgs_lse_f32 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/lse_trace.log 2>&1
python3 - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/lse_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def short(n): return re.sub(r'\(.*', '', n).replace('void ', '').replace('dzo::', '')[:40]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows]
# the timed stretch: the first 50 consecutive (gram_pass, reduce_finish, combine) triples after warm-up, no events
idx = [i for i, e in enumerate(ev) if e[2].startswith('combine_kernel')]
a, b = idx[20], idx[60]
seg = ev[a + 1:b + 1]
per = (ev[b][1] - ev[a][1]) / 1e3 / 40
busy = sum(e[1] - e[0] for e in seg) / 1e3 / 40
print('config 4 timeline: period per direction %.2f us, kernel busy %.2f us, gaps %.2f us' % (per, busy, per - busy))
t0 = seg[0][0]
for e in seg[:9]: print('%8.2f %7.2f  %s' % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2]))
PY
bash tools/collect_pmc_secondary.sh r03 > gpurun_out/r03_pmc2.log 2>&1; tail -25 gpurun_out/r03_pmc2.log
