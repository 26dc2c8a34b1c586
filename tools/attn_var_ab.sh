#!/bin/bash
# same-box A/B of attention library variants (built with -DDS_ATTN_VAR=n into build/libattn_var_n.so by tools/build_attn_vars.py)
for L in "96 16384" "192 4096"; do
  set -- $L
  for lib in build/libattn_var_*.so; do
    printf "%-28s " $lib
    DS_LIB=$(pwd)/$lib timeout -k 10 120 python tools/attn_microbench.py --c $1 --n $2 --batch 128 --iters 30 2>&1 | tail -1
  done
done
