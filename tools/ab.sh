#!/bin/bash
# Same-box A/B of library variants and / or environment switches (boxes of the pool differ by 3-5 %: never compare across gpurun calls).
#
#   bash tools/ab.sh [-m MODE] [-r ROUNDS] [-a "extra args"] VARIANT [VARIANT ...]
#
# VARIANT = [ENV=val[,ENV2=val2...]:][lib.so]     "libdiffusynth_hip_prev.so"   "DS_NO_CFG_PAIR=1:"   "DS_KSPLIT_FILL=512,DS_X=1:libfoo.so"
#           (no library name = the product build libdiffusynth_hip.so; libraries are file names under diffusynth_amd/, see build_variants.py)
# MODE    = bench  headline bench (default; -a e.g. "--dtype bf16", "--workload config2 --steps 20")
#           conv   conv_microbench.py over the representative 3x3 layers at U-Net batch 128 (-a e.g. "--tile 11 --iters 10"; LAYERS= overrides)
#           dw     dw_microbench.py over the eleven depthwise layer shapes (-a e.g. "--dtype fp32split")
#           attn   attn_microbench.py over the four attention levels (-a e.g. "--batch 128 --iters 20")
#           small  the small-batch workloads: BASELINE configs[1] (batch 16) and batch 1 (bench.py --workload config2 / config1)
# Rounds are interleaved (A B A B), so drift of the box shows up as disagreement between rounds.
MODE=bench; R=2; EXTRA=""
while getopts "m:r:a:" o; do case $o in m) MODE=$OPTARG;; r) R=$OPTARG;; a) EXTRA=$OPTARG;; *) exit 2;; esac; done
shift $((OPTIND - 1))
[ $# -ge 1 ] || { sed -n 2,15p "$0"; exit 2; }
JSON_LINE='import json,sys
d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}
print("%.1f steps/s  %.3f ms/step" % (d["value"], d["ms_per_step"]) + ("  dominant kernel %.1f us, frac %.4f" % (r["avg_launch_us"], r["frac"]) if r else ""))'
run_variant() {       # $1 = variant, rest = command
  local v=$1; shift
  local lib=${v##*:} envs=""
  [ "$v" != "$lib" ] && envs=$(echo "${v%:*}" | tr ',' ' ')
  [ -z "$lib" ] && lib=libdiffusynth_hip.so
  env $envs DS_LIB=$lib "$@"
}
CONV_LAYERS=${LAYERS:-"96:192:256:64:1:0 192:96:256:64:0:1 192:192:256:64:1:0 384:192:128:32:0:1 192:384:128:32:1:0 768:768:64:16:1:0 768:384:32:8:0:1"}
DW_LAYERS=${LAYERS:-"96:256:64 192:256:64 288:256:64 192:128:32 384:128:32 576:128:32 384:64:16 768:64:16 1152:64:16 768:32:8 1536:32:8"}
ATTN_LAYERS=${LAYERS:-"96:16384 192:4096 384:1024 384:256"}
for r in $(seq 1 $R); do
  case $MODE in
    bench)
      for v in "$@"; do printf "%-52s " "$v"
        run_variant "$v" timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 --warmup 2 $EXTRA 2>/dev/null | tail -1 | python3 -c "$JSON_LINE"; done;;
    small)
      for wl in config2 config1; do for v in "$@"; do printf "%-8s %-44s " $wl "$v"
        run_variant "$v" timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --workload $wl --steps 20 $EXTRA 2>/dev/null | tail -1 | python3 -c "$JSON_LINE"; done; done;;
    conv)
      for L in $CONV_LAYERS; do IFS=: read cin cout h w act res <<< "$L"; for v in "$@"; do printf "%-44s " "$v"
        run_variant "$v" timeout -k 10 120 python tools/conv_microbench.py --cin $cin --cout $cout --h $h --w $w --batch 128 --tile 11 --act $act --res $res --iters 10 $EXTRA 2>&1 | tail -1; done; done;;
    dw)
      for L in $DW_LAYERS; do IFS=: read c h w <<< "$L"; for v in "$@"; do printf "%-44s " "$v"
        run_variant "$v" timeout -k 10 120 python tools/dw_microbench.py --c $c --h $h --w $w --batch 128 --iters 20 $EXTRA 2>&1 | tail -1; done; done;;
    attn)
      for L in $ATTN_LAYERS; do IFS=: read c n <<< "$L"; for v in "$@"; do printf "%-44s " "$v"
        run_variant "$v" timeout -k 10 120 python tools/attn_microbench.py --c $c --n $n --batch 128 --iters 20 $EXTRA 2>&1 | tail -1; done; done;;
    *) echo "unknown mode $MODE"; exit 2;;
  esac
done
