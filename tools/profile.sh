#!/bin/bash
# tools/profile.sh <tag> -- rocprofv3 kernel-trace stats (+ separate PMC passes for HBM bytes) of the
# bench workload.  Outputs under gpurun_out/prof_<tag>/; copy the summaries you want judged to profiles/.
set -u
TAG=${1:-r01}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o ntt -- $CMD > "$OUT/stats.log" 2>&1
# PMC passes (own runs, no trace domains besides kernel-trace)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o ntt -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o ntt -- $CMD > "$OUT/pmc_write.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, os, statistics, sys
out = sys.argv[1]
def find(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None
summary = {}
st = find("stats/**/*kernel_stats.csv")
if st:
    summary["kernel_stats"] = list(csv.DictReader(open(st)))[:12]
# Per-dispatch view.  bench.py also launches the same kernels once on a 2-transform sample for its parity check;
# "full" = the dispatches with the largest grid of that kernel (the timed 256 x 2^20 workload).
tr = find("stats/**/*kernel_trace.csv")
if tr:
    per = {}
    for r in csv.DictReader(open(tr)):
        if "ntt_pass_kernel" not in r["Kernel_Name"]:
            continue
        per.setdefault(r["Kernel_Name"], []).append((int(r["Grid_Size_X"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    full = {}
    for k, v in per.items():
        g = max(x[0] for x in v)
        d = [x[1] for x in v if x[0] == g]
        full[k] = {"full_size_dispatches": len(d), "all_dispatches": len(v), "avg_ns": sum(d) / len(d), "median_ns": statistics.median(d),
                   "min_ns": min(d), "max_ns": max(d)}
    summary["ntt_pass_full_size_dispatches"] = full
    if full:
        summary["avg_launch_ms_over_both_passes"] = sum(v["avg_ns"] for v in full.values()) / len(full) / 1e6
for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(f"{name}/**/*counter_collection.csv")
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == counter and "ntt_pass_kernel" in r.get("Kernel_Name", "")]
    agg = {}
    for k in set(r["Kernel_Name"] for r in rows):
        g = max(int(r["Grid_Size"]) for r in rows if r["Kernel_Name"] == k)
        vals = [float(r["Counter_Value"]) for r in rows if r["Kernel_Name"] == k and int(r["Grid_Size"]) == g]
        agg[k] = {"full_size_dispatches": len(vals), "avg_KB_per_dispatch": sum(vals) / len(vals)}
    summary[counter] = agg
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:8000])
PY
