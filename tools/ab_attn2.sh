for v in "" attn1 attn2 attn4 attn8 attn16 attn31; do
  if [ -n "$v" ]; then export DS_LIB=libdiffusynth_hip_$v.so; else unset DS_LIB; fi
  echo -n "${v:-base}: "; timeout -k 10 120 python tools/attn_microbench.py --c 96 --n 16384 --batch 16 2>&1 | tail -1
  echo -n "${v:-base}: "; timeout -k 10 120 python tools/attn_microbench.py --c 192 --n 4096 --batch 16 2>&1 | tail -1
done
