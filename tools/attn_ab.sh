#!/bin/bash
# first- vs second-generation attention output pass (DS_ATTN_V1=1 selects the old kernels), per level at the headline U-Net batch
B=${1:-128}
for L in "96 16384" "192 4096" "384 1024" "384 256"; do
  set -- $L
  for v in 1 0; do
    if [ $v = 1 ]; then export DS_ATTN_V1=1; else unset DS_ATTN_V1; fi
    printf "v1=%s " $v
    timeout -k 10 120 python tools/attn_microbench.py --c $1 --n $2 --batch $B --iters 20 2>&1 | tail -1
  done
done
unset DS_ATTN_V1
