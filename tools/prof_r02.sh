#!/bin/bash
# tools/prof_r02.sh <tag> [target args...] -- rocprofv3 record of the BASELINE workloads (tools/prof_target.py):
#   stats   : --kernel-trace --stats (per-kernel durations)
#   pmc_sq  : GRBM_GUI_ACTIVE + SQ issue/wait counters (what binds the kernel; clock under load = GRBM_GUI_ACTIVE / 8 XCDs / duration)
#   pmc_mfma: SQ_INSTS_MFMA + SQ_VALU_MFMA_BUSY_CYCLES (the Tip5 kernels' v_mfma_f64_16x16x4_f64)
#   pmc_fetch / pmc_write : FETCH_SIZE, WRITE_SIZE in their own passes (MI355X_MICROARCH.md, HBM section)
# Each pass is its own run with --kernel-trace only.  Output: gpurun_out/prof_<tag>/summary.json (copy to profiles/).
set -u
TAG=${1:-r02}
shift || true
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export TF_PROF_IDENTITY=$OUT/library.json   # tools/prof_target.py writes the library's tf_version + source hash here
CMD=${TF_PROF_CMD:-"python $REPO/tools/prof_target.py $*"}   # TF_PROF_CMD: another target (tools/small_n_target.py)
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o t -- $CMD > "$OUT/stats.log" 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -o t -- $CMD > "$OUT/pmc_sq.log" 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM --output-format csv -d "$OUT/pmc_sq2" -o t -- $CMD > "$OUT/pmc_sq2.log" 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_WAVES --output-format csv -d "$OUT/pmc_mfma" -o t -- $CMD > "$OUT/pmc_mfma.log" 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o t -- $CMD > "$OUT/pmc_fetch.log" 2>&1
timeout -k 5 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o t -- $CMD > "$OUT/pmc_write.log" 2>&1
python3 "$REPO/tools/prof_summary.py" "$OUT" > "$OUT/summary.txt" 2>&1
tail -n 60 "$OUT/summary.txt"
