# usage (GPU box): bash scripts/all_ab.sh VARIANT...   all 22 layer times (scripts/layer_time.py, 16 views) of library variants built by mkvariant.sh (base = in-tree), one line per variant
for v in "$@"; do
  if [ "$v" != "base" ]; then export RNR_HIP_LIB=$PWD/build_abl/librnr_$v.so; else unset RNR_HIP_LIB; fi
  echo "== $v"; timeout 300 python scripts/layer_time.py --views 16 --winograd4 2>&1 | grep "^L\|rror\|fault\|^sum" | awk '{printf "%s %s ", $1, $(NF-6)} /^sum/ {print $0} END {print ""}'
done
