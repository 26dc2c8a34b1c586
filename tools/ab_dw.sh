for sh in "96 256 64" "192 256 64" "384 128 32" "768 64 16" "768 32 8"; do set -- $sh; timeout -k 10 120 python tools/dw_microbench.py --c $1 --h $2 --w $3 --batch ${B:-16} 2>&1 | tail -1; done
