# usage (GPU box): bash scripts/abl_wino_r04.sh VARIANT...   per-layer times of library variants (base = in-tree)
LAYERS=${LAYERS:-1,2,4,8,21}
for v in "$@"; do
  if [ "$v" != "base" ]; then export RNR_HIP_LIB=$PWD/build_abl/librnr_$v.so; else unset RNR_HIP_LIB; fi
  echo "== $v"; timeout 120 python scripts/layer_time.py --views ${VIEWS:-8} --winograd --layers $LAYERS 2>&1 | grep "^L\|rror\|fault" | head -30
done
