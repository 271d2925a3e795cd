#!/bin/bash
# tools/leaf_ab.sh -- leaf size of the zerofier tree (TF_TREE_LEAF_LOG = 7..10) against the batch evaluation at n = m = 2^14 .. 2^20,
# both fields; the defaults (256 points over BFE, 128 over XFE) come from this table (profiles/r02_leaf_ab.txt).
cd "$(dirname "$0")/.."
for w in 1 3; do
  for lg in 6 7 8 9 10; do
    echo "== width $w leaf 2^$lg"
    TF_TREE_LEAF_LOG=$lg python tools/batch_eval_sweep.py $w 2>&1 | grep -E "n 2\^14 m 2\^14|n 2\^16 m 2\^16|n 2\^18 m 2\^18|n 2\^20 m 2\^20"
  done
done
