#!/bin/bash
# tools/build_tip5_f64_variant.sh -- the product's sources with the Tip5 MDS on v_mfma_f64_16x16x4_f64 (-DTF_TIP5_I8=0, round 5's form, the
# yardstick of tools/microbench_mds.hip) -> twenty-first_amd/variants/libtf_hip_tip5f64.so (git-ignored and gpurun-ignored: built where it is needed; loaded through
# TF_HIP_LIBRARY; tools/switch_matrix.sh runs the whole GPU suite on it).  Only the Tip5 unit (and the ABI unit, which carries the hash) differ.
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
C=$REPO/twenty-first_amd/csrc
make -C "$C" -j8 >/dev/null
B=/tmp/tf_variant_tip5f64; mkdir -p "$B" "$REPO/twenty-first_amd/variants"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form -I$C"
/opt/rocm/bin/hipcc $COMMON -DTF_TIP5_I8=0 -DTF_SOURCE_HASH=\"variant-tip5f64\" -c -o "$B/tf_tip5.o" "$C/tf_tip5.hip" &
/opt/rocm/bin/hipcc $COMMON -DTF_TIP5_I8=0 -DTF_SOURCE_HASH=\"variant-tip5f64\" -c -o "$B/tf_abi.o" "$C/tf_abi.hip" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -no-hip-rt -Wl,--version-script=$C/tf_exports.map -o "$REPO/twenty-first_amd/variants/libtf_hip_tip5f64.so" \
  "$C/tf_ntt.o" "$B/tf_abi.o" "$C/tf_lat.o" "$B/tf_tip5.o" "$C/tf_poly.o" "$C/tf_multi.o"
echo "built twenty-first_amd/variants/libtf_hip_tip5f64.so"
