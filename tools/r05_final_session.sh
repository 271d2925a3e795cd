#!/bin/bash
# tools/r05_final_session.sh -- the closing GPU session of round 5 on the frozen sources: both suites, the C++ self-test, the bench
# line, the --force-dist records, the rocprofv3 kernel trace of the bench command, the PMC passes of the three BASELINE workloads,
# the reference bench shapes, the Tip5 microbenchmarks and level times.  Everything lands in gpurun_out/.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
{
  echo "== product library: python -m pytest tests -m gpu"
  python -m pytest tests -m gpu -q 2>&1 | tail -n 2
  echo "== laboratory library (TF_HIP_LIBRARY=libtf_hip_ab.so)"
  TF_HIP_LIBRARY=$REPO/twenty-first_amd/libtf_hip_ab.so python -m pytest tests -m gpu -q 2>&1 | tail -n 2
  echo "== C++ host mirror"
  twenty-first_amd/host/selftest 2>&1 | tail -n 9
  echo "== smoke"
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
} > gpurun_out/r05_final_suites.txt 2>&1
python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
echo "bench rc=$?" >> gpurun_out/r05_final_suites.txt
python bench.py --force-dist --config 5 2> /dev/null | tail -1 > gpurun_out/r05_c5_forcedist.json
python bench.py --force-dist --no-extra 2> /dev/null | tail -1 > gpurun_out/r05_c2_forcedist.json
timeout 900 bash tools/profile_bench.sh r05 > gpurun_out/r05_profile_bench.log 2>&1
timeout 1500 bash tools/prof_r02.sh r05p > gpurun_out/r05p_prof.log 2>&1
python tools/reference_bench_shapes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_reference_bench_shapes.txt
python tools/tip5_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_tip5_times.txt
python tools/tip5_small_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_tip5_small_times_final.txt
./tools/microbench_mds > gpurun_out/r05_microbench_mds_mfma.txt 2>&1
./tools/microbench_mfma_valu_mix > gpurun_out/r05_mfma_valu_mix.txt 2>&1
bash tools/merkle_trace.sh r05 24 > /dev/null 2>&1
python tools/ntt_sizes.py > gpurun_out/r05_ntt_sizes.txt 2>&1
cat gpurun_out/r05_final_suites.txt
