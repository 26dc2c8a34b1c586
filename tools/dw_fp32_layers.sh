for L in "96 256 64" "192 256 64" "288 256 64" "192 128 32" "384 128 32" "576 128 32" "384 64 16" "768 64 16" "1152 64 16" "768 32 8" "1536 32 8"; do
  set -- $L
  python tools/dw_microbench.py --c $1 --h $2 --w $3 --batch 128 --iters 10 --dtype fp32split 2>&1 | tail -1
done
