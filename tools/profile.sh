#!/bin/bash
# tools/profile.sh <tag> -- rocprofv3 kernel-trace stats (+ separate PMC passes for HBM bytes) of the
# bench workload.  Outputs under gpurun_out/prof_<tag>/; copy the summaries you want judged to profiles/.
set -u
TAG=${1:-r01}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o ntt -- $CMD > "$OUT/stats.log" 2>&1
# PMC passes (own runs, no trace domains besides kernel-trace)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o ntt -- $CMD > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o ntt -- $CMD > "$OUT/pmc_write.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
def find(pattern):
    g = glob.glob(os.path.join(out, pattern), recursive=True)
    return g[0] if g else None
summary = {}
st = find("stats/**/*kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    summary["kernel_stats"] = rows[:12]
for name, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(f"{name}/**/*counter_collection.csv")
    if not f:
        continue
    agg = {}
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = r.get("Kernel_Name", "?")
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r.get("Counter_Value", 0))
    summary[counter] = {k: {"dispatches": v[0], "sum": v[1], "avg_per_dispatch": v[1] / max(1, v[0])} for k, v in agg.items()}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:6000])
PY
