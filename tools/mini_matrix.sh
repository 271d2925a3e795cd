#!/bin/bash
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_switch_matrix_late.txt
: > $OUT
run() { res=$(env $1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 1); printf "%-34s %s\n" "$1" "$res" | tee -a $OUT; }
run "TF_DEFAULT=1"
run "TF_TREE_NO_FUSE=1"
run "TF_TREE_FUSE_INTERP=1"
run "TF_NTT_PERSIST=4"
run "TF_NTT_LAT2_NO_WIDE=1"
run "TF_NTT_LAT_MAX_WORDS=1073741824 TF_NTT_LAT2_MAX_WORDS=1073741824"
