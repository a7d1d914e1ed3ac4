for v in "$@"; do
  if [ "$v" != "base" ]; then export RNR_HIP_LIB=$PWD/build_abl/librnr_$v.so; else unset RNR_HIP_LIB; fi
  echo "== $v"; timeout 200 python scripts/layer_time.py --views 16 --winograd4 --layers 12,14,16,18,20 2>&1 | grep "^L\|rror\|fault"
done
