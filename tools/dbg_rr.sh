cd $GRAFT_REPO_ROOT
names=$(grep -n "^def test_" tests/test_gpu_parity.py | awk -F'[ (]' '{print $2}')
ids=""; j=0
for n in $names; do j=$((j+1)); if [ $j -ge 37 ] && [ $j -le 47 ]; then ids="$ids tests/test_gpu_parity.py::$n"; fi; done
TF_DEBUG_TIMING=1 python -m pytest $ids tests/test_gpu_parity.py::test_one_host_thread_round_robin_never_blocks -m gpu -q -s 2>&1 | grep -E "\[tf\] coset_eval_dev order 65536: hipFree|AssertionError|passed|failed" | cut -c1-300 | tail -30
