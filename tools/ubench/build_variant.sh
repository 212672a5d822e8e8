#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>: libd3feat_amd.so with gemm_f32.hip recompiled under the flags -> d3feat_amd/lib/variants/<name>.so
set -e
name=$1; shift
cd "$(dirname "$0")/../.."
mkdir -p d3feat_amd/lib/variants /tmp/d3f_variant_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=default "$@" -c d3feat_amd/csrc/gemm_f32.hip -o /tmp/d3f_variant_$name/gemm_f32.o
objs=$(ls d3feat_amd/lib/obj/*.o | grep -v gemm_f32.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o d3feat_amd/lib/variants/$name.so $objs /tmp/d3f_variant_$name/gemm_f32.o
echo built d3feat_amd/lib/variants/$name.so
