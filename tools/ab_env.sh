#!/bin/bash
# usage: ab_env.sh "NAME1:ENV1=V ENV2=V" "NAME2:" ... ; alternates bench runs (3 rounds) with the given environments
for r in 1 2 3; do
for kv in "$@"; do
  n=${kv%%:*}; e=${kv#*:}
  v=$(env $e python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "$r $n $v"
done; done
