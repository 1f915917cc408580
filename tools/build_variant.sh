#!/bin/bash
# an experiment build of the library: evalslide.hip recompiled with extra -D flags, linked with the product's other objects into
# tools/_build/libmprime_hip_<tag>.so (select it with MPRIME_LIBRARY=...).  usage: tools/build_variant.sh TAG [-DNAME=VALUE ...]
set -eu
cd "$(dirname "$0")/.."
tag=$1; shift
C=multiprime_amd/csrc
mkdir -p tools/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -pthread "$@" -c $C/evalslide.hip -o tools/_build/evalslide_$tag.o
objs=$(ls $C/_obj/*.o | grep -v evalslide.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o tools/_build/libmprime_hip_$tag.so tools/_build/evalslide_$tag.o $objs -ldl
echo tools/_build/libmprime_hip_$tag.so
