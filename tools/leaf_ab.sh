cd /root/repo
for w in 1 3; do for lg in 7 8 9 10; do echo "== width $w leaf 2^$lg"; TF_TREE_LEAF_LOG=$lg python tools/batch_eval_sweep.py $w 2>&1 | grep -v amdgpu.ids | grep -E "m 2\^(14|16|18|20)" | grep -E "n 2\^14 m 2\^14|n 2\^16 m 2\^16|n 2\^18 m 2\^18|n 2\^20 m 2\^20" ; done; done
