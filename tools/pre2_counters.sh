#!/bin/bash
# tools/pre2_counters.sh <tag> -- where do the parked wave-cycles of the configs[3] first pass go?  
# The two pass kernels of the 64 x 2^22 XFE coset evaluation (tools/prof_target.py --ntt 0 --merkle 0 --coset 3) under four
# separate --pmc passes: issue / wait split (SQ), vector-memory path (TA / TCP), L2 (TCC), L2 <-> fabric (TCC_EA).
# Output: gpurun_out/pre2_<tag>/summary.txt (per kernel, averages per dispatch).
set -u
TAG=${1:-r05}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pre2_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/prof_target.py --ntt 0 --merkle 0 --coset 3"
i=0
# (at most four TA / TCP / TCC counters per pass: more "exceeds the capabilities of the hardware to collect", and a failed pass of
# rocprofv3 does not exit by itself -- every pass runs under its own timeout)
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL" \
           "TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_STREAMING_REQ_sum"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/set$i" -o p -- $CMD > "$OUT/set$i.log" 2>&1 || echo "pass $i: rc $? ($SET)" >> "$OUT/failed_passes.txt"
done
python3 - "$OUT" > "$OUT/summary.txt" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "set*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "ntt_pass_kernel" not in k:
            continue
        a = agg[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for f in glob.glob(os.path.join(out, "set1/**/*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "ntt_pass_kernel" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(agg):
    d = dur.get(k, [])
    print(f"## {k.replace('void tfk::', '')[:110]}   dispatches {len(d)}  avg {sum(d) / max(1, len(d)):.1f} us (under the SQ counter pass)")
    c = {n: s / m for n, (m, s) in agg[k].items()}
    for n in sorted(c):
        print(f"   {n:44s} {c[n]:18.1f}")
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        print("   -- share of wave-cycles:  WAIT_ANY %.3f   WAIT_INST_ANY %.3f (of which LDS %.3f)   ACTIVE_INST_ANY %.3f   [VALU %.3f  VMEM %.3f  LDS %.3f]" % (
            c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_WAIT_INST_LDS", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            c.get("SQ_ACTIVE_INST_VALU", 0) / wc, c.get("SQ_ACTIVE_INST_VMEM", 0) / wc, c.get("SQ_ACTIVE_INST_LDS", 0) / wc))
    if c.get("TCC_REQ_sum"):
        print("   -- L2: hit rate %.3f   requests per dispatch %.3e   EA reads %.3e (32 B: %.3e)  EA writes %.3e" % (
            c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)), c["TCC_REQ_sum"], c.get("TCC_EA0_RDREQ_sum", 0), c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_EA0_WRREQ_sum", 0)))
    if c.get("TCP_TCC_READ_REQ_sum"):
        print("   -- TCP: avg read latency %.0f cycles (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ), cache accesses %.3e, reads to L2 %.3e, writes to L2 %.3e" % (
            c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / c["TCP_TCC_READ_REQ_sum"], c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0), c["TCP_TCC_READ_REQ_sum"], c.get("TCP_TCC_WRITE_REQ_sum", 0)))
PY
cat "$OUT/summary.txt"
