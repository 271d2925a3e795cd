#!/bin/bash
# tools/r06_final_session.sh -- the closing GPU session of round 6 on the frozen sources.  The PMC passes come FIRST and their records
# (profiles/valu_counts.json, hbm_traffic_ntt.json, pipeline_counters.json, stamped with this build's source hash) are written on the box, so
# the bench line that follows quotes library-matched counters.  Then: the suites, the C++ self-test, the bench line, the --force-dist records,
# the rocprofv3 kernel trace of the bench command, the A/B tables of this round's adoptions, sweeps, randomised parity and thread stress.
# A GPU call is limited to 60 minutes: `r06_final_session.sh a` = counters, records, suites, bench lines (what bench.py and DESIGN quote);
# `r06_final_session.sh b` = the A/B tables, sweeps, randomised parity, thread stress and the laboratory library's suite.  No argument: both.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
mkdir -p gpurun_out
T=r06
PART=${1:-ab}
if [[ $PART == *a* ]]; then
timeout 1500 bash tools/prof_r02.sh ${T}p > gpurun_out/${T}p_prof.log 2>&1
python tools/make_profile_records.py gpurun_out/prof_${T}p ${T}p > gpurun_out/${T}p_records.log 2>&1
timeout 900 bash tools/prof_r02.sh ${T}pipe --ntt 0 --merkle 0 --coset 0 --pipeline 3 > gpurun_out/${T}pipe_prof.log 2>&1
python tools/make_pipeline_record.py gpurun_out/prof_${T}pipe ${T}pipe 3 > gpurun_out/${T}pipe_records.log 2>&1
cp profiles/valu_counts.json profiles/hbm_traffic_ntt.json profiles/pipeline_counters.json gpurun_out/ 2>/dev/null
cp gpurun_out/prof_${T}p/summary.json gpurun_out/${T}p_rocprof_summary.json 2>/dev/null
cp gpurun_out/prof_${T}p/summary.txt gpurun_out/${T}p_rocprof_summary.txt 2>/dev/null
cp gpurun_out/prof_${T}pipe/summary.txt gpurun_out/${T}pipe_rocprof_summary.txt 2>/dev/null
cp gpurun_out/prof_${T}pipe/summary.json gpurun_out/${T}pipe_rocprof_summary.json 2>/dev/null
{
  echo "== product library: python -m pytest tests -m gpu"
  python -m pytest tests -m gpu -q 2>&1 | tail -n 2
  echo "== C++ host mirror"
  twenty-first_amd/host/selftest 2>&1 | tail -n 16
  echo "== smoke"
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
} > gpurun_out/${T}_final_suites.txt 2>&1
python bench.py > gpurun_out/${T}_bench_final.json 2> gpurun_out/${T}_bench_final.err
echo "bench rc=$?" >> gpurun_out/${T}_final_suites.txt
python bench.py --force-dist --config 5 2> /dev/null | tail -1 > gpurun_out/${T}_c5_forcedist.json
python bench.py --force-dist --no-extra 2> /dev/null | tail -1 > gpurun_out/${T}_c2_forcedist.json
timeout 900 bash tools/profile_bench.sh ${T} > gpurun_out/${T}_profile_bench.log 2>&1
cat gpurun_out/${T}_final_suites.txt
fi
if [[ $PART == *b* ]]; then
# configs[3] on ONE stream (its two kernels do not share the chip), under the counters
timeout 900 bash tools/prof_r02.sh ${T}c4 --ntt 0 --merkle 0 --coset 3 --pipe 1 > gpurun_out/${T}c4_prof.log 2>&1
cp gpurun_out/prof_${T}c4/summary.txt gpurun_out/${T}c4_rocprof_summary.txt 2>/dev/null
cp gpurun_out/prof_${T}c4/summary.json gpurun_out/${T}c4_rocprof_summary.json 2>/dev/null
python tools/c8_ab.py 10 --parity 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_c8_ab_final.txt
python tools/pipe_c5.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_pipe_tiles_final.txt
python tools/round_robin_latency.py 4 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_round_robin_latency_final.txt
python tools/reference_bench_shapes.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_reference_bench_shapes.txt
python tools/tip5_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_tip5_times.txt
python tools/tip5_small_times.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_tip5_small_times_final.txt
python tools/merkle_heights.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_merkle_heights_final.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I twenty-first_amd/csrc -o /tmp/microbench_mds tools/microbench_mds.hip && /tmp/microbench_mds > gpurun_out/${T}_microbench_mds_final.txt 2>&1
for h in 24 20 16; do bash tools/merkle_trace.sh ${T} $h > /dev/null 2>&1; done
python tools/ntt_sizes.py > gpurun_out/${T}_ntt_sizes.txt 2>&1
{
  for seed in 31 32 33; do timeout 400 python tools/fuzz_long.py $seed 150 2>&1 | grep -v amdgpu.ids | tail -n 1; done
  for seed in 41 42; do timeout 300 python tools/fuzz_long.py $seed 100 merkle,varlen,auth,trace 2>&1 | grep -v amdgpu.ids | tail -n 1; done
  timeout 200 python tools/stress_threads.py 40 20 2>&1 | grep -v amdgpu.ids | tail -n 1
  timeout 200 python tools/stress_ntt_threads.py 40 6 2>&1 | grep -v amdgpu.ids | tail -n 1
  timeout 200 python tools/stress_mixed_threads.py 40 6 2>&1 | grep -v amdgpu.ids | tail -n 1
} > gpurun_out/${T}_fuzz_long.txt 2>&1
# the laboratory library (every measured loser and diagnostic switch compiled in): built here, it does not travel
make -C twenty-first_amd/csrc ab -j8 > gpurun_out/${T}_make_ab.log 2>&1
{
  echo "== laboratory library (TF_HIP_LIBRARY=libtf_hip_ab.so)"
  TF_HIP_LIBRARY=$REPO/twenty-first_amd/libtf_hip_ab.so python -m pytest tests -m gpu -q 2>&1 | tail -n 2
} > gpurun_out/${T}_final_suites_lab.txt 2>&1
cat gpurun_out/${T}_final_suites_lab.txt
tail -n 8 gpurun_out/${T}_fuzz_long.txt
fi
