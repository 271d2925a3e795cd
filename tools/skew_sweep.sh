cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in base skew; do
  if [ $v = base ]; then unset TF_HIP_LIBRARY; else export TF_HIP_LIBRARY=$PWD/twenty-first_amd/variants/libtf_hip_$v.so; fi
  echo "== $v (rep $rep)"
  python tools/ntt_sizes.py 1 11 28 2>&1 | grep -v amdgpu
  python tools/ntt_sizes.py 3 11 26 2>&1 | grep -v amdgpu
done; done
unset TF_HIP_LIBRARY
echo "== pipe sweep on configs[3]"
for k in 1 2 3; do TF_NTT_PIPE=$k python tools/c8_variant_time.py -1 2>&1 | grep -v amdgpu; done
for mib in 512 1024; do TF_NTT_TILE_BYTES=$((mib<<20)) TF_NTT_PIPE=2 python tools/c8_variant_time.py -1 2>&1 | grep -v amdgpu; done
