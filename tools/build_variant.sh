#!/bin/bash
# Builds variants/lib_<name>.so = the product library with extra compiler flags (A/B experiments, stamp builds), in a scratch
# copy of csrc/ so the in-tree build is untouched.  Usage: bash tools/build_variant.sh <name> "<extra flags>"
# On the GPU box: cp variants/lib_<name>.so simgan_amd/libsimgan_hip.so (the box's copy of the tree is scratch).
set -eu
name=$1; extra=${2:-}
root=$(cd "$(dirname "$0")/.." && pwd)
work=/tmp/sg_variant_$name
rm -rf $work && mkdir -p $work/simgan_amd $root/variants
cp -r $root/simgan_amd/csrc $work/simgan_amd/ && cp -r $root/include $work/ && rm -rf $work/simgan_amd/csrc/build
make -s -C $work/simgan_amd/csrc -j8 ../libsimgan_hip.so CXXFLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-result -mllvm -amdgpu-kernarg-preload-count=16 $extra" 2>&1 | grep -E "error|warning" || true
cp $work/simgan_amd/libsimgan_hip.so $root/variants/lib_$name.so
ls -la $root/variants/lib_$name.so
