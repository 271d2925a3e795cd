#!/bin/bash
# tools/trace_oneshot.sh <tag> [log]: kernel trace of one-shot batch evaluation + interpolation (tools/oneshot_target.py), per-kernel table
set -u
TAG=${1:-r03}; LOG=${2:-12}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/${TAG}_oneshot_$LOG.txt
cd /tmp && export TMPDIR=/tmp
D=$REPO/gpurun_out/trace_${TAG}_oneshot$LOG
rm -rf "$D"
rocprofv3 --kernel-trace --output-format csv -d "$D" -o t -- python $REPO/tools/oneshot_target.py 1 $LOG > "$OUT" 2>&1
F=$(find "$D" -name '*kernel_trace.csv' | head -1)
python3 - "$F" >> "$OUT" <<'PY'
import csv,sys
rows=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
# last interpolation = after the last interpolant_unpad / direct... take the final fifth of the dispatches after the last leaf_evaluate
idx=max(i for i,(s,e,k) in enumerate(rows) if "leaf_evaluate" in k)
tail=rows[idx+1:]
n=len(tail)//5
one=tail[-n:]
print(f"last one-shot interpolation: {n} dispatches, span {(one[-1][1]-one[0][0])/1e3:.1f} us, kernels {sum(e-s for s,e,_ in one)/1e3:.1f} us")
for s,e,k in one:
    print(f"  {(s-one[0][0])/1e3:8.1f} +{(e-s)/1e3:6.2f} us  {k.split('(')[0].replace('void tfk::','')[:90]}")
PY
grep -v amdgpu.ids "$OUT"
