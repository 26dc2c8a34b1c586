# A/B of halo tile variants on representative layers (GPU box)
for t in 4 8; do for sh in "768 768 64 16" "384 384 128 32" "192 192 256 64" "96 192 256 64" "384 768 64 16"; do set -- $sh; echo -n "tile=$t cin=$1 cout=$2 ${3}x$4: "; timeout -k 10 120 python tools/conv_microbench.py --cin $1 --cout $2 --h $3 --w $4 --batch ${B:-16} --tile $t --iters 30 2>&1 | tail -1; done; done
