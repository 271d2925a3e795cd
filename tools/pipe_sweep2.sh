#!/bin/bash
# scratch-tile cache policy x tile size x streams: does the intermediate stay in the Infinity Cache when only the streamed sides are nt?
cd "$(dirname "$0")/.."
for lib in "" ab/libtf_hip_cs0ll0.so ab/libtf_hip_cs0ll2.so ab/libtf_hip_cs2ll0.so; do
  echo "== library ${lib:-default (nt everywhere)}"
  if [ -n "$lib" ]; then export TF_HIP_LIBRARY=$PWD/$lib; else unset TF_HIP_LIBRARY; fi
  python tools/pipe_sweep.py 20 2>&1 | grep -v amdgpu.ids | grep -E "2048 MiB|  256 MiB|  128 MiB|   64 MiB"
done
