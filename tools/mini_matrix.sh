#!/bin/bash
# tools/mini_matrix.sh -- the GPU suite under the switches that touch what changed AFTER the last full tools/switch_matrix.sh run
cd "$(dirname "$0")/.."
export TF_HIP_LIBRARY=${TF_HIP_LIBRARY:-$PWD/twenty-first_amd/libtf_hip_ab.so}  # the switches exist in the laboratory library only (csrc: make ab)
OUT=gpurun_out/r03_switch_matrix_late.txt
: > $OUT
run() { res=$(env $1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 1); printf "%-50s %s\n" "$1" "$res" | tee -a $OUT; }
run "TF_DEFAULT=1"
run "TF_TREE_NO_LEAF_SPLIT=1"
run "TF_TREE_NO_BUILD_LEVEL=1"
run "TF_TREE_NO_LEVEL=1 TF_TREE_NO_BUILD_LEVEL=1"
run "TF_NTT_NO_LAT=1"
run "TF_TREE_LEAF_LOG=6"
run "TF_TREE_INTERP_LEAF_LOG=8"
run "TF_TREE_LEVEL_XFE=1"
run "TF_NTT_LAT_MAX_WORDS=1073741824 TF_NTT_LAT2_MAX_WORDS=1073741824"
