#!/bin/bash
# tools/trace_oneshot_big.sh <tag> [log]: kernel trace of ONE-SHOT interpolation (tools/oneshot_target.py), the last call split into
# build / weights (walk down) / walk up, kernels grouped by name
set -u
TAG=${1:-r03}; LOG=${2:-20}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/${TAG}_oneshotbig_$LOG.txt
cd /tmp && export TMPDIR=/tmp
D=$REPO/gpurun_out/trace_${TAG}_oneshotbig$LOG
rm -rf "$D"
rocprofv3 --kernel-trace --output-format csv -d "$D" -o t -- python $REPO/tools/oneshot_target.py ${WIDTH:-1} $LOG > /dev/null 2>&1
F=$(find "$D" -name '*kernel_trace.csv' | head -1)
python3 - "$F" > "$OUT" <<'PY'
import csv,sys
rows=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
# timed loops come before the isolated calls; take the LAST call: from the last leaf_zerofier to the end
z=[i for i,(s,e,k) in enumerate(rows) if "leaf_zerofier" in k]
one=rows[z[-1]:]
d=next(i for i,(s,e,k) in enumerate(one) if "zerofier_derivative" in k)
u=next(i for i,(s,e,k) in enumerate(one) if "leaf_interpolant" in k)
def show(label, seg):
    span=(seg[-1][1]-seg[0][0])/1e3; busy=sum(e-s for s,e,_ in seg)/1e3
    print(f"## {label}: {len(seg)} dispatches, span {span:.1f} us, kernels {busy:.1f} us")
    by={}
    for s,e,k in seg:
        k=k.split('(')[0].replace('void tfk::','')[:100]
        by.setdefault(k,[0,0]); by[k][0]+=1; by[k][1]+=e-s
    for k,(c,t) in sorted(by.items(), key=lambda kv:-kv[1][1])[:14]:
        print(f"  {c:4d} x {t/c/1e3:9.2f} us = {t/1e3:9.1f} us  {k}")
print(f"# last one-shot interpolation: {len(one)} dispatches, span {(one[-1][1]-one[0][0])/1e3:.1f} us")
show("build", one[:d]); show("weights (derivative, walk down, inversion)", one[d:u]); show("walk up", one[u:])
PY
cat "$OUT"
