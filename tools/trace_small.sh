#!/bin/bash
# tools/trace_small.sh <tag>: kernel traces of the latency-bound shapes -- walks over a prepared zerofier tree (2^12 and 2^16 points)
# and single-slice transforms -- with the busy/idle split of tools/trace_gaps.py.  Output: gpurun_out/<tag>_small.txt
set -u
TAG=${1:-r03}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/${TAG}_small.txt
: > "$OUT"
cd /tmp && export TMPDIR=/tmp
for LOG in 12 16; do
  D=$REPO/gpurun_out/trace_${TAG}_tree$LOG
  rm -rf "$D"
  rocprofv3 --kernel-trace --output-format csv -d "$D" -o t -- python $REPO/tools/tree_walk_target.py 1 $LOG > /dev/null 2>&1
  F=$(find "$D" -name '*kernel_trace.csv' | head -1)
  echo "== prepared tree, 2^$LOG points, BFE: whole trace, then the last 5 interpolations only" >> "$OUT"
  python3 $REPO/tools/trace_gaps.py "$F" >> "$OUT"
  N=$(python3 - "$F" <<'PY'
import csv,sys
rows=sorted((int(r["Start_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
# dispatches after the last leaf_evaluate kernel = the 5 interpolations
idx=max(i for i,(s,k) in enumerate(rows) if "leaf_evaluate" in k)
print(len(rows)-idx-1)
PY
)
  python3 $REPO/tools/trace_gaps.py "$F" $N >> "$OUT"
done
echo "== single slices (HIP events over 200 back-to-back calls)" >> "$OUT"
python $REPO/tools/single_slice.py >> "$OUT" 2>&1
cat "$OUT"
