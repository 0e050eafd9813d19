#!/bin/bash
# A/B builds: tools/build_variant.sh <tag> <source.hip[,source2.hip...]> <extra flags...> -> variants/libmoe_hip_<tag>.so
# (the named translation units are recompiled with the extra flags; every other object is taken from cornell_moe_amd/build as it
#  is; select a variant at run time with MOE_LIB_PATH).
set -e
cd "$(dirname "$0")/.."
tag=$1; srcs=$2; shift 2
mkdir -p variants
objs=""; skip=""
for src in ${srcs//,/ }; do
  obj=variants/$(basename ${src%.hip})_$tag.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form "$@" \
    -Iinclude -c cornell_moe_amd/csrc/$src -o $obj &
  objs="$objs $obj"; skip="$skip|/$(basename ${src%.hip}).o"
done
wait
others=$(ls cornell_moe_amd/build/*.o | grep -v -E "${skip#|}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libmoe_hip_$tag.so $objs $others
ls -la variants/libmoe_hip_$tag.so
