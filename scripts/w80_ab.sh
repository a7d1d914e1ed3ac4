# usage (GPU box): bash scripts/w80_ab.sh VARIANT...   out-layer (L22) time of library variants (base = in-tree), 16 views
for v in "$@"; do
  if [ "$v" != "base" ]; then export RNR_HIP_LIB=$PWD/build_abl/librnr_$v.so; else unset RNR_HIP_LIB; fi
  echo "== $v"; timeout 200 python scripts/layer_time.py --views ${VIEWS:-16} --winograd4 --layers ${LAYERS:-22} 2>&1 | grep "^L\|rror\|fault"
done
