"""Dev tool: from a rocprofv3 --kernel-trace CSV of bench.py, the period of accepted single-pass steps
(pass start to next pass start) minus the pass's own duration = per-step overhead on the stream."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def short(n): return re.sub(r'\(.*', '', n).replace('void ', '').replace('dzo::', '')[:34]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])) for r in rows)
sp = [i for i, e in enumerate(ev) if e[2].startswith('lbfgs_single_pass')]
over, rej = [], []
for a, b in zip(sp[:-1], sp[1:]):
    names = [e[2] for e in ev[a + 1:b]]
    period = (ev[b][0] - ev[a][0]) / 1e3
    dur = (ev[a][1] - ev[a][0]) / 1e3
    (rej if any(n.startswith('trial_kernel') or n.startswith('gram_pass') for n in names) else over).append(period - dur)
print('accepted steps: %d, overhead per step (period - pass) avg %.1f us, min %.1f, max %.1f' % (len(over), sum(over) / len(over), min(over), max(over)))
if rej: print('steps with a rejected first trial: %d, extra per such step avg %.1f us' % (len(rej), sum(rej) / len(rej)))
a = sp[len(sp) // 2]
t0 = ev[a][0]
for e in ev[a:a + 12]: print('%9.1f %8.1f  %s' % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2]))
