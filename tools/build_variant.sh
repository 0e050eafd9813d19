#!/bin/bash
# A/B builds of one translation unit: tools/build_variant.sh <tag> <source.hip> <extra flags...> -> variants/libmoe_hip_<tag>.so
# (every other object is taken from cornell_moe_amd/build as it is; select a variant at run time with MOE_LIB_PATH).
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
mkdir -p variants
obj=variants/$(basename ${src%.hip})_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form "$@" \
  -Iinclude -c cornell_moe_amd/csrc/$src -o $obj
others=$(ls cornell_moe_amd/build/*.o | grep -v "/$(basename ${src%.hip}).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libmoe_hip_$tag.so $obj $others
ls -la variants/libmoe_hip_$tag.so
