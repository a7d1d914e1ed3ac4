run() { v=$1; L=$2; if [ "$v" != "base" ]; then export RNR_HIP_LIB=$PWD/build_abl/librnr_$v.so; else unset RNR_HIP_LIB; fi; echo "== $v"; timeout 300 python scripts/layer_time.py --views 16 --winograd4 --layers $L 2>&1 | grep "^L\|sum"; }
for v in base k1s2 k1s4 k1s5 base; do run $v 3,5,7,9; done
for v in base k2s7; do run $v 14,16,18,20; done
for v in base w80s3 w80b3 base; do run $v 22; done
