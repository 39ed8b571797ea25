#!/bin/bash
# Builds a variant of the library with extra compiler defines for the kernels (kernel experiments):
#   bash tools/build_variant.sh <name> -DRRT_PIPE_STEPS=4 ...   ->  rust-ray-tracer_amd/exp_libs/<name>.so
set -e
cd "$(dirname "$0")/../rust-ray-tracer_amd"
NAME=$1; shift
mkdir -p exp_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c csrc/rrt_kernels.hip -o exp_libs/$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o exp_libs/$NAME.so exp_libs/$NAME.o csrc/rrt_kernels_unrolled.o csrc/lbvh_build.o csrc/rrt_api.o csrc/scene_build.o csrc/bvh_builder.o host/gltf_loader.o host/png_writer.o -lpthread -ldl
rm -f exp_libs/$NAME.o
echo built exp_libs/$NAME.so
