#!/bin/bash
# usage: scripts/mkvariant.sh NAME "-DMACRO ..."   builds build_abl/librnr_NAME.so = the in-tree library with conv.hip recompiled with
# the given macros (CPU side, cross-compile); select it at run time with RNR_HIP_LIB=$PWD/build_abl/librnr_NAME.so
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/.."
C=relightable-nr_amd/csrc
mkdir -p build_abl
make -C $C -s -j4
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $FLAGS -c $C/conv.hip -o build_abl/conv_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/build/common.o $C/build/raster.o $C/build/raster_bwd.o $C/build/textures.o \
    $C/build/shade.o $C/build/objparse.o build_abl/conv_$NAME.o -o build_abl/librnr_$NAME.so
echo built build_abl/librnr_$NAME.so
