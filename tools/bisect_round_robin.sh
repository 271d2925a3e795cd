cd $GRAFT_REPO_ROOT
names=$(grep -n "^def test_" tests/test_gpu_parity.py | awk -F'[ (]' '{print $2}')
i=0
for grp in "1 12" "13 24" "25 36" "37 47"; do
  set -- $grp
  ids=""
  j=0
  for n in $names; do j=$((j+1)); if [ $j -ge $1 ] && [ $j -le $2 ]; then ids="$ids tests/test_gpu_parity.py::$n"; fi; done
  echo "== group $1..$2"
  python -m pytest $ids tests/test_gpu_parity.py::test_one_host_thread_round_robin_never_blocks -m gpu -q -p no:randomly 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-300
done
