for sh in "96 16384" "192 4096" "384 1024" "384 256"; do set -- $sh; timeout -k 10 120 python tools/attn_microbench.py --c $1 --n $2 --batch ${B:-16} 2>&1 | tail -1; done
