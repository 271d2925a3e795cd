#!/bin/bash
# tools/profile_bench.sh <tag> -- rocprofv3 --kernel-trace --stats of the bench command itself (NTT leg), so that the per-kernel
# average durations and bench.py's HIP-event launch time come from the SAME run.  Output: gpurun_out/profbench_<tag>/.
set -u
TAG=${1:-r02}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/profbench_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o b -- python $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
python3 - "$OUT" <<'PY'
import csv, glob, json, os, statistics, sys
out = sys.argv[1]
bench = json.load(open(os.path.join(out, "bench_under_rocprof.json")))
tr = glob.glob(os.path.join(out, "stats/**/*kernel_trace.csv"), recursive=True)[0]
per = {}
for r in csv.DictReader(open(tr)):
    if "ntt_pass_kernel" in r["Kernel_Name"]:
        per.setdefault(r["Kernel_Name"], []).append((int(r["Grid_Size_X"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
res = {}
for k, v in per.items():
    g = max(x[0] for x in v)
    full = sorted([x for x in v if x[0] == g], key=lambda x: x[1])
    d = [x[2] - x[1] for x in full]
    last = d[-60:]  # the timed region (50 steps) and the 10 steps after it: the last 60 full-size dispatches
    res[k] = {"full_size_dispatches": len(d), "avg_us_all": sum(d) / len(d) / 1e3, "avg_us_last_60": sum(last) / len(last) / 1e3,
              "median_us_last_60": statistics.median(last) / 1e3}
summary = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra",
           "kernels": res,
           "avg_launch_ms_from_trace_last_60": sum(v["avg_us_last_60"] for v in res.values()) / len(res) / 1e3,
           "bench_under_rocprof": {"ms_per_step": bench["ms_per_step"], "avg_launch_ms_hip_events": bench["roofline"]["avg_launch_ms"],
                                   "frac": bench["roofline"]["frac"], "value": bench["value"]}}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
