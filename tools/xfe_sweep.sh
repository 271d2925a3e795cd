#!/bin/bash
# XFE split sweep around the R = 1024 last pass (word-granular tiles): default plan vs forced splits
for n in 23 24 25 26; do
  echo "default n=$n"; timeout 100 python tools/ntt_sizes.py 3 $n $n 2>&1 | grep width | cut -c1-62
  r=$((n-10)); a0=$(((r+1)/2)); a1=$((r-a0)); echo "3-pass $a0,$a1,10 n=$n"; TF_NTT_EXPERIMENT=1 TF_NTT_SPLIT3=$a0,$a1 timeout 100 python tools/ntt_sizes.py 3 $n $n 2>&1 | grep width | cut -c1-62
done
for n in 15 16; do
  echo "default n=$n"; timeout 100 python tools/ntt_sizes.py 3 $n $n 2>&1 | grep width | cut -c1-62
  echo "2-pass last=10 n=$n"; TF_NTT_EXPERIMENT=1 TF_NTT_SPLIT2=$((n-10)) timeout 100 python tools/ntt_sizes.py 3 $n $n 2>&1 | grep width | cut -c1-62
done
