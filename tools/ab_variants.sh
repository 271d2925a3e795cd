#!/bin/bash
# tools/ab_variants.sh <outdir> <rounds> "name|ENV=.. ENV=..|lib" ...   -- short headline bench (NTT leg only) of several library
# builds / environment settings, interleaved `rounds` times on the same box so that drift shows up in every variant alike.
# lib = path relative to the repo (empty: the shipped twenty-first_amd/libtf_hip.so).
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/$1; shift
ROUNDS=$1; shift
mkdir -p "$OUT"
cd "$REPO"
for r in $(seq 1 "$ROUNDS"); do
  for spec in "$@"; do
    name=${spec%%|*}; rest=${spec#*|}; envs=${rest%%|*}; lib=${rest#*|}
    libenv=""; [ -n "$lib" ] && libenv="TF_HIP_LIBRARY=$REPO/$lib"
    env $envs $libenv timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra > "$OUT/${name}_$r.json" 2> "$OUT/${name}_$r.err"
    python - "$OUT/${name}_$r.json" "$name" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(f"{sys.argv[2]:28s} {d['ms_per_step']:.4f} ms  frac {d['roofline']['frac']:.4f}  step_ms_after {d['step_ms_after']['min']:.4f}..{d['step_ms_after']['max']:.4f}  sclk {d['sclk_mhz']['after_timed_region']:.0f}")
except Exception as e:
    print(f"{sys.argv[2]:28s} FAILED {e}")
PY
  done
done
