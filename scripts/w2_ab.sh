# usage (GPU box): bash scripts/w2_ab.sh VARIANT...   F(2x2,2x2) layer times (the ten 4x4 stride-2 layers) of library variants (base = in-tree)
for v in "$@"; do
  if [ "$v" != "base" ]; then export RNR_HIP_LIB=$PWD/build_abl/librnr_$v.so; else unset RNR_HIP_LIB; fi
  echo "== $v"; timeout 300 python scripts/layer_time.py --views ${VIEWS:-16} --winograd4 --layers ${LAYERS:-3,5,7,9,11,12,14,16,18,20} 2>&1 | grep "^L\|rror\|fault\|SUM\|sum" | head -30
done
