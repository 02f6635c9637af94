cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
rm -rf gpurun_out/tl_1e6; mkdir -p gpurun_out/tl_1e6
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_1e6 -o t -- python3 bench.py --dim 1000000 --steps 100 --warmup 5 --no-cpu-baseline --no-kernel-events --no-two-pass > gpurun_out/tl_1e6/log.txt 2>&1
python3 - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/tl_1e6/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n): return re.sub(r'[<(].*', '', n).replace('void ', '').replace('dzo::', '')[:34]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows)
sp = [i for i, e in enumerate(ev) if e[2].startswith('lbfgs_point_pass')]
a = sp[len(sp) * 3 // 4]
t0 = ev[a][0]
for e in ev[a:a + 16]: print('%9.2f start %8.2f dur  %s' % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2]))
per=[(ev[b][0]-ev[a_][0])/1e3 for a_,b in zip(sp[:-1],sp[1:])][-80:]
print('period between passes (last 80): avg %.2f min %.2f' % (sum(per)/len(per), min(per)))
PY
