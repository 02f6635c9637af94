# GPU box: SQ occupancy / stall counters of the headline pass (one --pmc pass for SQ, one for GRBM).
#   bash tools/sq_counters.sh <tag> [kernel substring] [bench args...]
set -e
TAG="${1:-sq}"; KERN="${2:-lbfgs_point_pass_kernel<double, 20, false>}"; shift || true; shift || true
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/sq_$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events --no-two-pass $@"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/sq -o s -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/grbm -o g -- $CMD > $OUT/grbm.log 2>&1
python3 - "$OUT" "$KERN" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, kern = sys.argv[1], sys.argv[2]
res = {}
for sub in ("sq", "grbm"):
    f = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        print("no counter file for", sub); continue
    agg = defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    t = glob.glob(os.path.join(out, sub, "**", "*kernel_trace.csv"), recursive=True)
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(t[0])) if kern in r["Kernel_Name"]] if t else []
    for k, v in agg.items():
        tail = v[-30:]
        res[k] = sum(tail) / len(tail)
    if durs:
        res[f"dur_us_{sub}"] = sum(durs[-30:]) / len(durs[-30:]) / 1e3
for k, v in sorted(res.items()):
    print(f"{k:24s} {v:16.1f}")
w = res.get("SQ_WAVE_CYCLES")
if w:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
        if k in res: print(f"{k}/SQ_WAVE_CYCLES = {res[k] / w:.3f}")
if "GRBM_GUI_ACTIVE" in res and "dur_us_grbm" in res:
    print(f"effective clock ~ {res['GRBM_GUI_ACTIVE'] / 8 / res['dur_us_grbm'] / 1e3:.3f} GHz (GRBM_GUI_ACTIVE / 8 / duration)")
PY
