#!/bin/bash
# tools/profile_sizes.sh <width> <lo> <hi> -- per-pass kernel durations (rocprofv3 kernel trace) of tools/ntt_sizes.py
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_sizes
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o s -- python $REPO/tools/ntt_sizes.py "$@" > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**/*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# group consecutive identical (kernel, grid) sequences: print the pattern of each timed loop once
seq = [(r["Kernel_Name"].replace("tfk::", "").replace("(tfk::NttPassArgs)", "")[:60], int(r["Grid_Size_X"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if "ntt" in r["Kernel_Name"]]
agg = {}
order = []
for k, g, d in seq:
    if (k, g) not in agg:
        agg[(k, g)] = []
        order.append((k, g))
    agg[(k, g)].append(d)
for k, g in order:
    v = agg[(k, g)]
    print(f"{k:62s} grid {g:9d}  n {len(v):3d}  avg {sum(v)/len(v)/1e3:8.1f} us  min {min(v)/1e3:8.1f}")
PY
grep width "$OUT/log.txt" | cut -c1-70
