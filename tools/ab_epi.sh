# where does the time of the big-spatial 3x3 layers go?  (GPU box; needs `tools/ablate.sh 16 32 48` libs)
for sh in "192 192 256 64" "96 192 256 64" "768 768 64 16"; do set -- $sh
  for v in base noact nofold nostats abl16 abl32 abl48; do
    unset DS_LIB; extra=""
    case $v in noact) extra="--act 0";; nofold) extra="--fold 0";; nostats) extra="--stats 0";; abl*) export DS_LIB="libdiffusynth_hip_${v}.so";; esac
    echo -n "$v cin=$1 cout=$2 ${3}x$4: "; timeout -k 10 120 python tools/conv_microbench.py --cin $1 --cout $2 --h $3 --w $4 --batch 16 --tile 4 --iters 30 $extra 2>&1 | tail -1
  done
done
