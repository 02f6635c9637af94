"""Turn rocprofv3 CSV output (kernel trace + separate --pmc passes) into the summaries that
are committed under profiles/.

    python tools/summarize_profile.py <round-tag> <kernel_trace_dir> [<pmc_fetch_dir> <pmc_write_dir>]

HBM bytes per launch follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane)
coalesced streaming read, so the read side is doubled; WRITE_SIZE is exact for 16-B stores.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("dzo::", "")
    return n.strip()


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return hits[0] if hits else None


def kernel_stats(trace_dir, steady=50):
    f = find(trace_dir, "*kernel_trace.csv")
    rows = list(csv.DictReader(open(f)))
    agg = defaultdict(list)
    for r in rows:
        agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = []
    total = sum(sum(v) for v in agg.values())
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        # the bench's timed region is its LAST `steps` iterations (history full, k = m); the
        # earlier launches belong to the untimed ring fill / warm-up and stream fewer vectors
        tail = v[-steady:] if len(v) >= steady else v
        out.append({"kernel": k, "calls": len(v), "total_us": round(sum(v) / 1e3, 1), "avg_us": round(sum(v) / len(v) / 1e3, 2),
                    "min_us": round(min(v) / 1e3, 2), "max_us": round(max(v) / 1e3, 2), "pct": round(100 * sum(v) / total, 2),
                    f"timed_region_avg_us_last{steady}": round(sum(tail) / len(tail) / 1e3, 2)})
    return out


def pmc(pmc_dir, counter):
    f = find(pmc_dir, "*counter_collection.csv")
    agg = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg


def main():
    tag, trace_dir = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    stats = kernel_stats(trace_dir)
    with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
        w.writeheader()
        w.writerows(stats)
    print(f"wrote profiles/{tag}_kernel_stats.csv ({len(stats)} kernels)")
    if len(sys.argv) >= 5:
        fetch, write = pmc(sys.argv[3], "FETCH_SIZE"), pmc(sys.argv[4], "WRITE_SIZE")
        out = {}
        for k in sorted(set(fetch) | set(write)):
            fv, wv = fetch.get(k, []), write.get(k, [])
            # the full-history launches are the largest ones: report the upper quartile mean
            def top(v):
                v = sorted(v)
                t = v[len(v) // 2:] if v else []
                return sum(t) / len(t) if t else 0.0
            rd = 2.0 * top(fv) * 1024.0
            wr = top(wv) * 1024.0
            out[k] = {"launches_fetch_pass": len(fv), "launches_write_pass": len(wv),
                      "FETCH_SIZE_KiB_avg": round(top(fv), 1), "WRITE_SIZE_KiB_avg": round(top(wv), 1),
                      "read_bytes_per_launch": int(rd), "write_bytes_per_launch": int(wr),
                      "hbm_bytes_per_launch": int(rd + wr),
                      "note": "read = 2 x FETCH_SIZE x 1024 (gfx950 correction), write = WRITE_SIZE x 1024; mean over the larger half of launches"}
        for long_name in list(out):
            if long_name.startswith("gram_pass_lanes_kernel<double"):
                out["lbfgs_gram_pass"] = out[long_name]
            if long_name.startswith("combine_kernel<double"):
                out["lbfgs_combine"] = out[long_name]
            if long_name.startswith("lbfgs_single_pass_kernel<double") and "lbfgs_single_pass" not in out:
                out["lbfgs_single_pass"] = out[long_name]
            if long_name.startswith("lbfgs_point_pass_kernel<double") and not long_name.startswith("lbfgs_point_pass_kernel<double, 8, true"):
                # the default optimizer's pass = bench.py's profiling label "lbfgs_single_pass" (`<double, 8, true, ...>` is the first step's)
                out["lbfgs_single_pass"] = out[long_name]
        json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
        json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
        print(f"wrote profiles/{tag}_pmc.json")


if __name__ == "__main__":
    main()
