#!/bin/bash
# tools/merkle_trace.sh <tag> [log]: kernel trace of tools/merkle_trace_target.py; every kernel of the LAST Merkle build by launch
# (grid, duration) and the flat hash_pairs / permute launches
set -u
TAG=${1:-r05}; LOG=${2:-24}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/${TAG}_merkle_trace_$LOG.txt
cd /tmp && export TMPDIR=/tmp
D=$REPO/gpurun_out/trace_${TAG}_merkle$LOG
rm -rf "$D"
rocprofv3 --kernel-trace --output-format csv -d "$D" -o t -- python $REPO/tools/merkle_trace_target.py $LOG > /dev/null 2>&1
F=$(find "$D" -name '*kernel_trace.csv' | head -1)
python3 - "$F" > "$OUT" <<'PY'
import csv,sys
rows=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"],int(r["Grid_Size_X"]),int(r["Workgroup_Size_X"])) for r in csv.DictReader(open(sys.argv[1]))))
rows=[r for r in rows if "tip5" in r[2] or "merkle" in r[2]]
tops=[i for i,r in enumerate(rows) if "merkle_top" in r[2]]
seg=rows[tops[-2]+1:tops[-1]+1]
print(f"## last Merkle build: {len(seg)} launches, span {(seg[-1][1]-seg[0][0])/1e3:.1f} us, kernels {sum(e-s for s,e,*_ in seg)/1e3:.1f} us")
for s,e,k,g,w in seg:
    print(f"  {(e-s)/1e3:10.2f} us  grid {g//w:8d} x {w:4d}  {k.split('(')[0].replace('void tfk::','')[:90]}")
print("## after the builds")
for s,e,k,g,w in rows[tops[-1]+1:]:
    print(f"  {(e-s)/1e3:10.2f} us  grid {g//w:8d} x {w:4d}  {k.split('(')[0].replace('void tfk::','')[:90]}")
PY
cat "$OUT"
