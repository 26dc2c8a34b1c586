#!/bin/bash
# first- vs second-generation attention kernels per level at the headline U-Net batch:
#   DS_ATTN_V1=1 selects the old kernels for both passes, DS_ATTN_CTX1=1 only the old context pass
B=${1:-128}
for L in "96 16384" "192 4096" "384 1024" "384 256"; do
  set -- $L
  for v in V1 CTX1 none; do
    unset DS_ATTN_V1 DS_ATTN_CTX1
    [ $v = V1 ] && export DS_ATTN_V1=1
    [ $v = CTX1 ] && export DS_ATTN_CTX1=1
    printf "old=%s " $v
    timeout -k 10 120 python tools/attn_microbench.py --c $1 --n $2 --batch $B --iters 20 2>&1 | tail -1
  done
done
unset DS_ATTN_V1 DS_ATTN_CTX1
