#!/bin/bash
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
tools/ab_variants.sh r02b 2 \
  "base||" \
  "nolazy||ab/libtf_hip_nolazy.so" \
  "bflycc||ab/libtf_hip_bflycc.so" \
  "montcc||ab/libtf_hip_montcc.so" \
  "ldstw||ab/libtf_hip_ldstw.so" \
  "w4_generic_r4096|TF_NTT_NO_LAST1024=1 TF_NTT_ROUND_ELEMS=4096|" \
  "w6_generic_r4096|TF_NTT_NO_LAST1024=1 TF_NTT_ROUND_ELEMS=4096|ab/libtf_hip_w6.so" \
  "w4_generic|TF_NTT_NO_LAST1024=1|" \
  "w6_default||ab/libtf_hip_w6.so" \
  "nt3|TF_NTT_NT=3|" \
  "nt3_pipe2_128|TF_NTT_NT=3 TF_NTT_PIPE=2 TF_NTT_TILE_BYTES=134217728|"
