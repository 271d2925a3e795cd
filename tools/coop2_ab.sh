#!/bin/bash
# tools/coop2_ab.sh -- the two-row (32-lane) permutation of small Tip5 launches against the 16-lane form it replaces there: the laboratory
# library with and without TF_TIP5_NO_COOP2, same box, back to back: tree builds by height, the Tip5 / Merkle rows of the reference's bench
# shapes, small hash_varlen / hash_pair launches.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
[ -f twenty-first_amd/libtf_hip_ab.so ] || make -C twenty-first_amd/csrc ab -j16 > /dev/null 2>&1
export TF_HIP_LIBRARY=$REPO/twenty-first_amd/libtf_hip_ab.so
for mode in two_rows sixteen_lanes two_rows sixteen_lanes; do
  if [ $mode = sixteen_lanes ]; then export TF_TIP5_NO_COOP2=1; else unset TF_TIP5_NO_COOP2; fi
  echo "== $mode (TF_TIP5_NO_COOP2=${TF_TIP5_NO_COOP2:-unset})"
  python tools/merkle_heights.py 2>&1 | grep -v amdgpu.ids | grep -E "height +(4|8|10|12|14|16|18|20):"
  python tools/reference_bench_shapes.py 2>&1 | grep -v amdgpu.ids | grep -E "^hash_|^merkle_"
  python tools/tip5_small_times.py 2>&1 | grep -v amdgpu.ids | head -12
done
