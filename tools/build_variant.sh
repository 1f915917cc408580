#!/bin/bash
# an experiment build of the library: ONE translation unit of csrc/ recompiled with extra -D flags, linked with the product's other
# objects into tools/_build/libmprime_hip_<tag>.so (select it with MPRIME_LIBRARY=...).
# usage: tools/build_variant.sh TAG SOURCE [-DNAME=VALUE ...]        e.g. tools/build_variant.sh s8 unique.hip -DMP_HIST_FLUSH_S=8
set -eu
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
C=multiprime_amd/csrc
mkdir -p tools/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -pthread -Iinclude "$@" -c $C/$src -o tools/_build/${src%.*}_$tag.o
objs=$(ls $C/_obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o tools/_build/libmprime_hip_$tag.so tools/_build/${src%.*}_$tag.o $objs -ldl
echo tools/_build/libmprime_hip_$tag.so
