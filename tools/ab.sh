#!/bin/bash
# usage: ab.sh name1=path1 name2=path2 ... ; alternates runs, 3 rounds
for r in 1 2 3; do
for kv in "$@"; do
  n=${kv%%=*}; p=${kv#*=}
  v=$(TF_HIP_ALLOW_OLDER_LIBRARY=1 TF_HIP_LIBRARY=$p python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('parity'))")
  echo "$r $n $v"
done; done
