#!/bin/bash
# tools/switch_matrix.sh [outfile] -- the whole GPU suite under every A/B switch in turn (each alternative path must be bit-exact
# too), then the long randomised parity run on the product build.  The switches exist in the LABORATORY library only
# (csrc: make ab -> twenty-first_amd/libtf_hip_ab.so, selected through TF_HIP_LIBRARY); the product library ignores them.
# First line of the record: the product library with no switch.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$REPO/gpurun_out/switch_matrix.txt}
mkdir -p "$(dirname "$OUT")"
cd "$REPO"
: > "$OUT"
ROW=0
run() {  # <env assignment> <description>
  local res
  ROW=$((ROW + 1))
  # TF_MATRIX_FROM / TF_MATRIX_TO: only rows FROM..TO (1-based) -- a GPU call has a time limit, the matrix is longer than one call
  if [ "$ROW" -lt "${TF_MATRIX_FROM:-1}" ] || [ "$ROW" -gt "${TF_MATRIX_TO:-9999}" ]; then return; fi
  res=$(env $1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 1)
  printf "%-34s %-100s %s\n" "$1" "($2)" "$res" | tee -a "$OUT"
}
run "TF_DEFAULT=1" "no switch: the shipped plan, PRODUCT library"
export TF_HIP_LIBRARY=${TF_HIP_LIBRARY:-$REPO/twenty-first_amd/libtf_hip_ab.so}
[ -f "$TF_HIP_LIBRARY" ] || { echo "missing $TF_HIP_LIBRARY: make -C twenty-first_amd/csrc ab" | tee -a "$OUT"; exit 1; }
run "TF_DEFAULT=1" "no switch: the shipped plan, laboratory library"
run "TF_NTT_TILE_BYTES=33554432" "32 MiB scratch slabs: every batched multi-pass transform runs in many tiles"
run "TF_NTT_TILE_BYTES=33554432 TF_NTT_PIPE=3" "... dealt to three side streams"
run "TF_NTT_NO_LAST1024=1" "generic kernel instead of the R = 1024 last-pass specialisation"
run "TF_NTT_NO_R1024=1" "run-time-P2 column pass (COL) instead of the constant-P2 one"
run "TF_NTT_NO_COL=1" "generic kernel (canonical networks, plain pointers) for the column passes other than R = 1024"
run "TF_NTT_NO_R1024=1 TF_NTT_NO_COL=1" "generic kernel for every column pass"
run "TF_NTT_NO_GFAST=1" "single-pass transforms with the column-major thread order"
run "TF_COSET_EVAL_NO_SPLIT=1" "blown-up coset evaluation by zero padding instead of interleaved cosets"
run "TF_POLY_MUL_NO_FUSE=1" "fast_multiply with separate pad / Hadamard / truncate passes"
run "TF_NTT_NO_BLOCK=1" "two-pass plans instead of the whole-transform-per-workgroup kernel, 2^11..2^14"
run "TF_NTT_NO_XFE_BLOCK=1" "XFE slices of 2^11 / 2^12 points through two-pass plans instead of the block kernel's limb transforms"
run "TF_NTT_NO_ROWS32=1" "ntt_tiny_kernel / the row pass instead of the wave-private tile kernel for transforms of at most 64 points"
run "TF_NTT_ROWS32_WG=1" "round 2's workgroup-tile kernel for batches of 32-point transforms (round 5: laboratory only)"
run "TF_NTT_NO_WORDS16=1" "XFE rows through the last passes as tiles of whole elements, not whole cache lines"
run "TF_NTT_NO_COL_SHIFT=1" "last-pass tile boundaries not shifted to the output's cache-line alignment"
run "TF_NTT_NO_SCALED_LAST1024=1" "generic last pass for fast_coset_interpolate instead of the R = 1024 kernel's scaled tail"
run "TF_NTT_NO_SMALL_LAUNCH=1" "512-thread tiles and the R = 1024 last pass for small calls too"
run "TF_NTT_WG_THREADS=256" "256-thread workgroups everywhere"
run "TF_NTT_NT=3" "non-temporal loads / stores in the generic kernels as well"
run "TF_NTT_NO_PRE2=1" "2^21 / 2^22 points in three passes instead of two (round 3)"
run "TF_NTT_NO_LAT=1" "no latency-shaped kernels: small calls on the pass / block kernels (round 3)"
run "TF_NTT_NO_LAT2=1" "latency-shaped kernel for 64..4096 points only, not the two-pass latency plan (round 3)"
run "TF_NTT_LAT_MAX_WORDS=1073741824 TF_NTT_LAT2_MAX_WORDS=1073741824" "latency-shaped kernels for every call they can serve, whatever its size"
run "TF_TREE_INTERP_LEAF_LOG=8" "trees that are walked upwards with 256-point leaves as in round 2"
run "TF_TREE_NO_FUSE=1" "tree walks with the reverse / remainder steps as kernels of their own (round 3)"
run "TF_TREE_NO_LEVEL=1" "tree walks with a level's transforms as launches of their own instead of one launch per level (round 3)"
run "TF_TREE_NO_LEVEL=1 TF_TREE_NO_FUSE=1" "... and the elementwise steps as kernels too"
run "TF_TREE_NO_BUILD_LEVEL=1" "tree build with a level's transforms as launches of their own (round 3)"
run "TF_TREE_LEVEL_XFE=1" "one launch per level over XFieldElement too (three thread groups per line; measured loss, off by default)"
run "TF_TREE_NO_LEAF_SPLIT=1" "one thread per point / per coefficient in the leaf kernels of small walks too (round 3)"
run "TF_TREE_FUSE_INTERP=1" "the interpolation's pointwise combination fused into the inverse transform's load (round 3, off by default)"
run "TF_NTT_PERSIST=4" "the R = 1024 column pass as chains of four tiles per workgroup (round 3, off by default)"
run "TF_NTT_LAT2_NO_WIDE=1" "256-thread slices for the 1024-point lines of the latency plan (round 3)"
run "TF_BATCH_EVAL=tree" "zerofier tree wherever it applies"
run "TF_BATCH_EVAL=horner" "Horner everywhere"
run "TF_BATCH_EVAL=tree TF_TREE_UNIT_SLAB=3000" "zerofier tree with the units of a walk cut into slabs of a few units"
run "TF_TREE_LEAF_LOG=6" "64-point leaves in the zerofier tree (deeper trees)"
run "TF_TREE_LEAF_LOG=10" "1024-point leaves"
run "TF_TIP5_NO_COOP2=1" "16 lanes per permutation in every small Tip5 launch and subtree level (round 6 puts a permutation on a row pair where rows idle)"
run "TF_NTT_PRE4=1" "2^22-point coset evaluations as 1024 x 4096 with the radix-4 last pass (round 4, measured loss)"
run "TF_NTT_PIPE=1" "batch tiles of every multi-pass plan on ONE stream (round 6 deals the 2^21 / 2^22 two-pass tiles to two)"
run "TF_NTT_PIPE=4 TF_NTT_TILE_BYTES=268435456" "256 MiB tiles on four side streams"
unset TF_HIP_LIBRARY
# round 6: the f64 form of the Tip5 MDS (round 5's product, -DTF_TIP5_I8=0) is still a buildable variant; built where hipcc is
# (tools/build_tip5_f64_variant.sh, built on the box when missing): the whole suite on it
V=$REPO/twenty-first_amd/variants/libtf_hip_tip5f64.so
[ -f "$V" ] || bash "$REPO/tools/build_tip5_f64_variant.sh" > /dev/null 2>&1   # (variants/ does not travel to the GPU box: built where it is needed)
if [ -f "$V" ]; then
  export TF_HIP_LIBRARY=$V
  run "TF_DEFAULT=1" "variant library: Tip5 MDS on v_mfma_f64_16x16x4_f64 (-DTF_TIP5_I8=0), everything else the product"
  unset TF_HIP_LIBRARY
fi
[ -n "${TF_MATRIX_NO_FUZZ:-}" ] && exit 0
echo "--- long randomised parity run on the product library (tools/fuzz_long.py, 3 seeds x 120 s)" | tee -a "$OUT"
for seed in 11 12 13; do
  timeout 400 python tools/fuzz_long.py $seed 120 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee -a "$OUT"
done
