#!/bin/bash
# tools/trace_big_tree.sh <tag> [log]: kernel trace of walks over a PREPARED tree of 2^log points (tools/tree_walk_target.py): the last
# evaluation and the last interpolation, kernels grouped by name with launch counts and total time
set -u
TAG=${1:-r03}; LOG=${2:-20}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/${TAG}_bigtree_$LOG.txt
cd /tmp && export TMPDIR=/tmp
D=$REPO/gpurun_out/trace_${TAG}_bigtree$LOG
rm -rf "$D"
rocprofv3 --kernel-trace --output-format csv -d "$D" -o t -- python $REPO/tools/tree_walk_target.py 1 $LOG > /dev/null 2>&1
F=$(find "$D" -name '*kernel_trace.csv' | head -1)
python3 - "$F" > "$OUT" <<'PY'
import csv,sys
rows=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
ev=[i for i,(s,e,k) in enumerate(rows) if "leaf_evaluate" in k]
li=[i for i,(s,e,k) in enumerate(rows) if "leaf_interpolant" in k]
def show(label, seg):
    span=(seg[-1][1]-seg[0][0])/1e3; busy=sum(e-s for s,e,_ in seg)/1e3
    print(f"## {label}: {len(seg)} dispatches, span {span:.1f} us, kernels {busy:.1f} us")
    by={}
    for s,e,k in seg:
        k=k.split('(')[0].replace('void tfk::','')[:100]
        by.setdefault(k,[0,0]); by[k][0]+=1; by[k][1]+=e-s
    for k,(c,t) in sorted(by.items(), key=lambda kv:-kv[1][1]):
        print(f"  {c:4d} x {t/c/1e3:9.2f} us = {t/1e3:9.1f} us  {k}")
# last evaluation: from after the previous leaf_evaluate to the last leaf_evaluate before the first leaf_interpolant of the final block
last_ev=max(i for i in ev if i < li[-5])
prev_ev=max(i for i in ev if i < last_ev)
show("last prepared-tree evaluation", rows[prev_ev+1:last_ev+1])
show("last prepared-tree interpolation", rows[li[-1]:])
PY
cat "$OUT"
