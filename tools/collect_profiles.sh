# GPU box: everything that is committed under profiles/ for one round.
#   bash tools/collect_profiles.sh r02
# rocprofv3 passes are separate (kernel trace / stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE), see profiles/README.md.
set -e
TAG="${1:-r02}"
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-kernel-events"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o f -- $CMD > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o w -- $CMD > $OUT/write.log 2>&1
echo "write done"
python3 tools/summarize_profile.py $TAG $OUT/trace $OUT/fetch $OUT/write
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) profiles/${TAG}_rocprofv3_kernel_stats_raw.csv
python3 - "$TAG" <<'PY'
# the two-pass leg's kernels (Gram pass, combine, reduce, finish) with their PMC traffic, as one small table
import csv, json, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(f'profiles/{tag}_kernel_stats.csv')))
pmc = json.load(open(f'profiles/{tag}_pmc.json'))
keep = [r for r in rows if r['kernel'].startswith('gram_pass_lanes_kernel<double, true, 4, false>') or r['kernel'].startswith('combine_kernel<double')
        or r['kernel'] in ('gram_reduce_kernel', 'gram_finish_kernel')]
with open(f'profiles/{tag}_two_pass_kernel_stats.csv', 'w') as f:
    w = csv.DictWriter(f, fieldnames=list(keep[0].keys()) + ['pmc_read_bytes_per_launch', 'pmc_write_bytes_per_launch'])
    w.writeheader()
    for r in keep:
        p = pmc.get(r['kernel'], {})
        r['pmc_read_bytes_per_launch'] = p.get('read_bytes_per_launch', ''); r['pmc_write_bytes_per_launch'] = p.get('write_bytes_per_launch', '')
        w.writerow(r)
PY
bash tools/sq_counters.sh ${TAG} "lbfgs_point_pass_kernel<double, 20, false, 1, false, 0>" > profiles/${TAG}_sq_counters_pass.txt 2>&1 || true
python3 bench.py > profiles/${TAG}_bench_line.json 2> $OUT/bench.err
echo "bench line done"
for w in bfgs_dense bfgs_batched lbfgs_lse_f32; do python3 bench.py --workload $w > profiles/${TAG}_bench_$w.json 2> $OUT/bench_$w.err; echo "$w done"; done
python3 bench.py --workload adgd --steps 200 --warmup 10 > profiles/${TAG}_bench_adgd.json 2> $OUT/bench_adgd.err
BENCH_DIST_BACKEND=gloo BENCH_FORCE_DEVICE=0 python3 bench.py --gpus 2 --steps 20 --warmup 3 2> $OUT/bench_gloo.err | grep '^{' > profiles/${TAG}_bench_gloo_rehearsal_2ranks_1gpu.json || true   # (gloo prints its connection banner on stdout)
mkdir -p gpurun_out/profiles_$TAG; cp profiles/${TAG}_* profiles/pmc_latest.json gpurun_out/profiles_$TAG/
ls -la gpurun_out/profiles_$TAG
