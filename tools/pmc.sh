#!/bin/bash
# tools/pmc.sh <tag> -- SQ counter passes for the NTT pass kernel (separate runs, kernel-trace only).
set -u
TAG=${1:-x}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra"
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/set$i" -o p -- $CMD > "$OUT/set$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(out, "set*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "ntt_pass" not in r["Kernel_Name"]:
            continue
        a = agg[r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    n, s = agg[k]
    print(f"{k:28s} dispatches {n:5d}  avg/dispatch {s/n:16.1f}")
PY
