cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
names=$(grep -n "^def test_" tests/test_gpu_parity.py | awk -F'[ (]' '{print $2}')
ids=""; j=0
for n in $names; do j=$((j+1)); if [ $j -ge 43 ] && [ $j -le 45 ]; then ids="$ids tests/test_gpu_parity.py::$n"; fi; done
cd /tmp
rocprofv3 --hip-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rr_trace -o t -- python -m pytest --rootdir $GRAFT_REPO_ROOT $(for i in $ids; do echo $GRAFT_REPO_ROOT/$i; done) $GRAFT_REPO_ROOT/tests/test_gpu_parity.py::test_one_host_thread_round_robin_never_blocks -m gpu -q 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-300
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/rr_trace -name "*hip_api_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "hip api calls; columns", list(rows[0].keys()))
long = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], int(r["Start_Timestamp"])) for r in rows]
t_end = max(x[2] for x in long)
tail = [x for x in long if x[2] > t_end - 200_000_000]   # the last 0.2 s
from collections import Counter
c = Counter(); tot = Counter()
for d, f, _ in tail:
    c[f] += 1; tot[f] += d
for f, t in tot.most_common(12):
    print(f"{f:40s} calls {c[f]:6d} total {t/1e3:10.1f} us  mean {t/c[f]/1e3:8.1f} us")
print("longest calls in the last 0.2 s:", sorted(tail, reverse=True)[:12])
PY
rm -rf gpurun_out/rr_trace
