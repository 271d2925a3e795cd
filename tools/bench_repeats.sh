#!/bin/bash
# tools/bench_repeats.sh [runs] [outfile] -- python bench.py (the default line: every leg, CPU legs included in the first run only) N times back
# to back on one box: the run-to-run spread of every leg, one line per run, from each line's `summary` block.  The first run's full line is
# kept beside the table (gpurun_out/<tag>_bench_final.json).
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
cd "$REPO"
N=${1:-5}
OUT=${2:-gpurun_out/r06_bench_repeats.txt}
mkdir -p "$(dirname "$OUT")"
H=$(python -c "import twenty_first_amd as tf; print(tf.lib().tf_source_hash().decode())" 2>/dev/null)
echo "# python bench.py, $N runs back to back on one MI355X, library $H (run 1 with the CPU legs, the others --no-cpu-baseline): run-to-run spread of every leg" > "$OUT"
for i in $(seq 1 "$N"); do
  if [ "$i" = 1 ]; then python bench.py > gpurun_out/_rep.json 2> /dev/null; cp gpurun_out/_rep.json gpurun_out/r06_bench_final.json
  else python bench.py --no-cpu-baseline > gpurun_out/_rep.json 2> /dev/null; fi
  python - >> "$OUT" <<'PY'
import json
d = json.loads(open("gpurun_out/_rep.json").read().strip().splitlines()[-1])
s = d["summary"]
print(f"headline GFelts/s {s['headline']['value']} ms {s['headline']['ms']} frac {s['headline']['frac']} | merkle G leaves/s {s['merkle']['value'] / 1e9:.3f} ms {s['merkle']['ms']} | "
      f"coset ms {s['coset_eval']['ms']} frac {s['coset_eval']['frac']} | config5 GFelts/s {s['config5']['value']} trees G leaves/s {s['config5']['trees_leaves_per_s'] / 1e9:.3f} | commit ms {s['commit_pipeline']['ms']}")
PY
done
rm -f gpurun_out/_rep.json
cat "$OUT"
