#!/bin/bash
# usage: scripts/mkvariant_shade.sh NAME "-DMACRO ..."   like mkvariant.sh, for shade.hip
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/.."
C=relightable-nr_amd/csrc
mkdir -p build_abl
make -C $C -s -j4
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off $FLAGS -c $C/shade.hip -o build_abl/shade_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $C/build/common.o $C/build/raster.o $C/build/raster_bwd.o $C/build/textures.o \
    build_abl/shade_$NAME.o $C/build/objparse.o $C/build/conv.o -o build_abl/librnr_$NAME.so
echo built build_abl/librnr_$NAME.so
