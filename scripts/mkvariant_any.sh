#!/bin/bash
# usage: scripts/mkvariant_any.sh FILE NAME "-DMACRO ..."   builds build_abl/librnr_NAME.so = the in-tree library with csrc/FILE.hip
# (conv | raster | shade) recompiled with the given macros; select it at run time with RNR_HIP_LIB=$PWD/build_abl/librnr_NAME.so
set -e
FILE=$1; NAME=$2; FLAGS=$3
cd "$(dirname "$0")/.."
C=relightable-nr_amd/csrc
mkdir -p build_abl
make -C $C -s -j4
if [ "$FILE" = conv ]; then EXTRA=-fno-slp-vectorize; else EXTRA=-ffp-contract=off; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $EXTRA $FLAGS -c $C/$FILE.hip -o build_abl/${FILE}_$NAME.o
OBJS=""
for o in common raster raster_bwd textures shade objparse conv; do
    if [ "$o" = "$FILE" ]; then OBJS="$OBJS build_abl/${FILE}_$NAME.o"; else OBJS="$OBJS $C/build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o build_abl/librnr_$NAME.so
echo built build_abl/librnr_$NAME.so
