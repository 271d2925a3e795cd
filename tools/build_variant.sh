#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG=.. ..." -- an experimental build of the NTT unit linked with the product's other objects:
#   twenty-first_amd/variants/libtf_hip_NAME.so  (git-ignored; travels to the GPU box; loaded through TF_HIP_LIBRARY)
set -eu
NAME=$1; FLAGS=${2:-}
REPO=$(cd "$(dirname "$0")/.." && pwd)
C=$REPO/twenty-first_amd/csrc
make -C "$C" -j8 >/dev/null
B=/tmp/tf_variant_$NAME; mkdir -p "$B"
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-result -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form -I$C"
/opt/rocm/bin/hipcc $COMMON $FLAGS -DTF_SOURCE_HASH=\"variant-$NAME\" -c -o "$B/tf_ntt.o" "$C/tf_ntt.hip" &
/opt/rocm/bin/hipcc $COMMON $FLAGS -DTF_SOURCE_HASH=\"variant-$NAME\" -c -o "$B/tf_abi.o" "$C/tf_abi.hip" &
wait
mkdir -p "$REPO/twenty-first_amd/variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -no-hip-rt -Wl,--version-script=$C/tf_exports.map -o "$REPO/twenty-first_amd/variants/libtf_hip_$NAME.so" \
  "$B/tf_ntt.o" "$B/tf_abi.o" "$C/tf_lat.o" "$C/tf_tip5.o" "$C/tf_poly.o" "$C/tf_multi.o"
echo "built twenty-first_amd/variants/libtf_hip_$NAME.so"
