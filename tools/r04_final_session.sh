#!/bin/bash
# tools/r04_final_session.sh -- the closing GPU session of round 4 on the frozen sources: both suites, the bench line, the rocprofv3
# kernel trace of the bench command, the PMC passes of the three BASELINE workloads.  Everything lands in gpurun_out/.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
{
  echo "== product library: python -m pytest tests -m gpu"
  python -m pytest tests -m gpu -q 2>&1 | tail -n 2
  echo "== laboratory library (TF_HIP_LIBRARY=libtf_hip_ab.so)"
  TF_HIP_LIBRARY=$REPO/twenty-first_amd/libtf_hip_ab.so python -m pytest tests -m gpu -q 2>&1 | tail -n 2
  echo "== laboratory library, TF_NTT_PRE4=1 (the radix-4 last pass as the automatic plan of fast_coset_evaluate at 2^22)"
  TF_HIP_LIBRARY=$REPO/twenty-first_amd/libtf_hip_ab.so TF_NTT_PRE4=1 python -m pytest tests -m gpu -q 2>&1 | tail -n 2
  echo "== C++ host mirror"
  twenty-first_amd/host/selftest 2>&1 | tail -n 6
} > gpurun_out/r04_final_suites.txt 2>&1
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err
echo "bench rc=$?" >> gpurun_out/r04_final_suites.txt
bash tools/profile_bench.sh r04 > gpurun_out/r04_profile_bench.log 2>&1
bash tools/prof_r02.sh r04p > gpurun_out/r04p_prof.log 2>&1
cat gpurun_out/r04_final_suites.txt
