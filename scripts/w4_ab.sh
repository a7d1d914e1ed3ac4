# usage (GPU box): bash scripts/w4_ab.sh VARIANT...   F(4x4,3x3) layer times of library variants (base = in-tree)
for v in "$@"; do
  if [ "$v" != "base" ]; then export RNR_HIP_LIB=$PWD/build_abl/librnr_$v.so; else unset RNR_HIP_LIB; fi
  echo "== $v"; timeout 120 python scripts/layer_time.py --views ${VIEWS:-8} --winograd4 --layers ${LAYERS:-1,2,4,6,8} 2>&1 | grep "^L\|rror\|fault" | head -30
done
