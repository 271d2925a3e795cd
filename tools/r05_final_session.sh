#!/bin/bash
# tools/r05_final_session.sh -- the closing GPU session of round 5 on the frozen sources.  Order matters: the PMC passes come FIRST and
# their records (profiles/valu_counts.json, hbm_traffic_ntt.json, stamped with this build's source hash) are written on the box, so the
# bench line that follows quotes library-matched counters.  Then: both suites, the C++ self-test, the bench line, the --force-dist
# records, the rocprofv3 kernel trace of the bench command, the reference bench shapes, the Tip5 / Merkle timings and traces, the size
# sweep, randomised parity and thread stress.  Everything lands in gpurun_out/ (the two json records are copied there as well).
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
mkdir -p gpurun_out
timeout 1500 bash tools/prof_r02.sh r05p > gpurun_out/r05p_prof.log 2>&1
python tools/make_profile_records.py gpurun_out/prof_r05p r05p > gpurun_out/r05p_records.log 2>&1
cp profiles/valu_counts.json profiles/hbm_traffic_ntt.json gpurun_out/ 2>/dev/null
cp gpurun_out/prof_r05p/summary.json gpurun_out/r05p_rocprof_summary.json 2>/dev/null
cp gpurun_out/prof_r05p/summary.txt gpurun_out/r05p_rocprof_summary.txt 2>/dev/null
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
python tools/reference_bench_shapes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_reference_bench_shapes.txt
python tools/tip5_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_tip5_times.txt
python tools/tip5_small_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_tip5_small_times_final.txt
python tools/merkle_heights.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_merkle_heights_final.txt
./tools/microbench_mds > gpurun_out/r05_microbench_mds_mfma.txt 2>&1
./tools/microbench_mfma_valu_mix > gpurun_out/r05_mfma_valu_mix.txt 2>&1
for h in 24 20 16; do bash tools/merkle_trace.sh r05 $h > /dev/null 2>&1; done
python tools/ntt_sizes.py > gpurun_out/r05_ntt_sizes.txt 2>&1
{
  for seed in 31 32 33; do timeout 400 python tools/fuzz_long.py $seed 150 2>&1 | grep -v amdgpu.ids | tail -n 1; done
  for seed in 41 42; do timeout 300 python tools/fuzz_long.py $seed 100 merkle,varlen,auth,trace 2>&1 | grep -v amdgpu.ids | tail -n 1; done
  timeout 200 python tools/stress_threads.py 40 20 2>&1 | grep -v amdgpu.ids | tail -n 1
  timeout 200 python tools/stress_ntt_threads.py 40 6 2>&1 | grep -v amdgpu.ids | tail -n 1
  timeout 200 python tools/stress_mixed_threads.py 40 6 2>&1 | grep -v amdgpu.ids | tail -n 1
} > gpurun_out/r05_fuzz_long.txt 2>&1
cat gpurun_out/r05_final_suites.txt
tail -n 8 gpurun_out/r05_fuzz_long.txt
