#!/bin/bash
# A/B builds of libvitk.so: recompiles the named source with extra -D flags and links it with the in-tree objects of the rest.
# usage: tools/build_variant.sh NAME SOURCE.hip "-DFLAG=1 ..."   ->  ab/libvitk_NAME.so   (select with VITK_LIB_PATH)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; flags=$3
csrc=$root/vision_transformers_torch_xla_amd/csrc
mkdir -p $root/ab /tmp/vitk_ab_$name
obj=/tmp/vitk_ab_$name/${src%.hip}.o
extra=""; [ "$src" = gemm.hip -o "$src" = gemm_ntq.hip ] && extra="-mllvm -amdgpu-atomic-optimizer-strategy=None"      # as build.py's EXTRA_FLAGS
[ "$src" = attention.hip ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function $extra $flags -Wno-inline-asm -Wno-cuda-compat -Wno-pass-failed -c $csrc/$src -o $obj
others=$(ls $csrc/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/ab/libvitk_$name.so $obj $others
echo $root/ab/libvitk_$name.so
