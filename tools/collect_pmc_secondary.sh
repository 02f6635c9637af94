# GPU box: FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, --kernel-trace only) and rocprofv3 kernel times of the
# secondary workloads' dominant kernels -> profiles/<tag>_pmc_secondary.md
#   bash tools/collect_pmc_secondary.sh r03
set -e
TAG="${1:-r03}"
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc2_$TAG
rm -rf $OUT; mkdir -p $OUT
run() {  # name, bench args...
  local name=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${name}_$C -o p -- python3 bench.py "$@" --no-cpu-baseline > $OUT/${name}_$C.log 2>&1 || echo "pass $name $C failed"
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_trace -o t -- python3 bench.py "$@" --no-cpu-baseline > $OUT/${name}_trace.log 2>&1 || echo "trace $name failed"
}
run dense --workload bfgs_dense --steps 20 --warmup 3
run batched --workload bfgs_batched --steps 40 --warmup 4 --poll 10
run adgd --workload adgd --steps 60 --warmup 5
run lse --workload lbfgs_lse_f32 --steps 50 --warmup 5
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
def short(n): return re.sub(r'\(.*', '', n).replace('void ', '').replace('dzo::', '').strip()
def pmc(d, c):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    agg = defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            if r['Counter_Name'] == c: agg[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return agg
def durs(d):
    f = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
    agg = defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])): agg[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return agg
want = {'dense': ('tri_pass_kernel', 'tri_reduce_kernel', 'quadratic_phi6_kernel', 'quadratic_phi6_cols_kernel', 'bfgs_move', 'finish_phi6_advance_kernel', 'norm2_pair_begin_kernel'), 'batched': ('batch_step_kernel',),
        'adgd': ('adgd_fused_rosen_kernel',), 'lse': ('gram_pass_lanes_kernel', 'gram_reduce_finish_kernel', 'combine_kernel')}
table = {}
lines = [f'# PMC traffic and rocprofv3 kernel times of the secondary workloads ({tag})', '',
         'Separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes (`--kernel-trace` only) and one `--kernel-trace --stats` pass per',
         'workload (`tools/collect_pmc_secondary.sh`); read = 2 x FETCH_SIZE x 1024 (gfx950 correction, MI355X_MICROARCH.md), write =',
         'WRITE_SIZE x 1024; mean over the second half of the launches (the timed region). The counters sit on the L2\'s fabric side:',
         'reads served by the 256 MiB Infinity Cache are included.', '',
         '| workload | kernel | launches | avg us (trace pass) | read / launch | write / launch |', '|---|---|---|---|---|---|']
for name, pats in want.items():
    fe, wr, du = pmc(os.path.join(out, name + '_FETCH_SIZE'), 'FETCH_SIZE'), pmc(os.path.join(out, name + '_WRITE_SIZE'), 'WRITE_SIZE'), durs(os.path.join(out, name + '_trace'))
    for k in sorted(set(fe) | set(du)):
        if not any(p in k for p in pats): continue
        half = lambda v: v[len(v) // 2:] if v else []
        m = lambda v: sum(half(v)) / len(half(v)) if half(v) else float('nan')
        lines.append(f'| {name} | `{k}` | {len(du.get(k, []))} | {m(du.get(k, [])):.2f} | {2 * m(fe.get(k, [])) * 1024 / 1e6:.2f} MB | {m(wr.get(k, [])) * 1024 / 1e6:.2f} MB |')
        table.setdefault(name, {})[k] = {'launches': len(du.get(k, [])), 'avg_us': round(m(du.get(k, [])), 2),
                                          'read_bytes_per_launch': round(2 * m(fe.get(k, [])) * 1024), 'write_bytes_per_launch': round(m(wr.get(k, [])) * 1024)}
open(f'profiles/{tag}_pmc_secondary.md', 'w').write('\n'.join(lines) + '\n')
table['_source'] = f'profiles/{tag}_pmc_secondary.md (tools/collect_pmc_secondary.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction)'
json.dump(table, open('profiles/pmc_secondary_latest.json', 'w'), indent=1)
print('\n'.join(lines))
PY
mkdir -p gpurun_out/profiles_$TAG; cp profiles/${TAG}_pmc_secondary.md profiles/pmc_secondary_latest.json gpurun_out/profiles_$TAG/
