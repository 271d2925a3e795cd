#!/bin/bash
# Round-2 GPU session A: validate the new arithmetic blocks, A/B lazy networks, sweep tile size x tile streams, full bench,
# GPU test suite, rocprof record.  Everything lands in gpurun_out/r02a/.
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/r02a
mkdir -p "$OUT"
cd "$REPO"
export PYTHONUNBUFFERED=1
echo "== microbench blocks"; timeout 300 tools/microbench_blocks > "$OUT/microbench_blocks.txt" 2>&1; tail -n 45 "$OUT/microbench_blocks.txt"
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; tail -n 3 "$OUT/smoke.txt"
echo "== quick parity subset"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or config2" > "$OUT/pytest_quick.txt" 2>&1; tail -n 5 "$OUT/pytest_quick.txt"
echo "== A/B lazy vs canonical networks (short bench, NTT leg only)"
for i in 1 2; do
  timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra > "$OUT/bench_lazy_$i.json" 2>"$OUT/bench_lazy_$i.err"; python -c "import json,sys; d=json.load(open('$OUT/bench_lazy_$i.json')); print('lazy   ', d['ms_per_step'], d['roofline']['frac'], d['sclk_mhz'])"
  TF_HIP_LIBRARY=$REPO/ab/libtf_hip_nolazy.so timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra > "$OUT/bench_nolazy_$i.json" 2>"$OUT/bench_nolazy_$i.err"; python -c "import json,sys; d=json.load(open('$OUT/bench_nolazy_$i.json')); print('nolazy ', d['ms_per_step'], d['roofline']['frac'], d['sclk_mhz'])"
done
echo "== tile x stream sweep"; timeout 900 python tools/pipe_sweep.py 30 > "$OUT/pipe_sweep.txt" 2>&1; cat "$OUT/pipe_sweep.txt"
echo "== full bench"; timeout 1200 python bench.py > "$OUT/bench_full.json" 2>"$OUT/bench_full.err"; tail -c 3000 "$OUT/bench_full.json"; tail -n 5 "$OUT/bench_full.err"
echo "== counters list"; (cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "mall|hbm|dram|umc|EA_|TCC_EA|TCC_REQ|TCC_HIT|TCC_MISS" | head -80) > "$OUT/counters_list.txt" 2>&1; wc -l "$OUT/counters_list.txt"
echo "== rocprof record"; timeout 1500 tools/prof_r02.sh r02a > "$OUT/prof.txt" 2>&1; tail -n 70 "$OUT/prof.txt"
echo "== full GPU suite"; timeout 2400 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.txt" 2>&1; tail -n 8 "$OUT/pytest_gpu.txt"
