#!/bin/bash
# Builds the library with extra compiler flags into tools/scratch/lib_<name>.so (git-ignored; travels to the GPU box),
# e.g.  tools/build_variant.sh scanprof -DBROTLI_AMD_PROFILE_SCAN ; then BROTLI_AMD_LIB=tools/scratch/lib_scanprof.so python bench.py ...
set -eu
REPO=$(cd "$(dirname "$0")/.." && pwd)
PKG=$REPO/rust-brotli-decompressor_amd
NAME=$1; shift
D=$REPO/tools/scratch/build_$NAME
mkdir -p "$D"
FLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fvisibility=hidden -I$REPO/include -I$PKG/csrc $*"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -c "$PKG/csrc/brotli_kernels.hip" -o "$D/k.o" &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -DBROTLI_AMD_GANG_KERNEL -c "$PKG/csrc/brotli_kernels.hip" -o "$D/kg.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -c "$PKG/csrc/brotli_capi.cpp" -o "$D/c.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$REPO/tools/scratch/lib_$NAME.so" "$D/k.o" "$D/kg.o" "$D/c.o" "$PKG/csrc/dict_blob.o"
echo "built tools/scratch/lib_$NAME.so"
