#!/bin/bash
# Builds a variant of the library with extra compiler options for the kernels (kernel experiments):
#   bash tools/build_variant.sh <name> -DRRT_PIPE_STEPS=4 ...   ->  rust-ray-tracer_amd/exp_libs/<name>.so
# The options go to both translation units of the render kernel; UNROLLED_ONLY=1 applies them to the one for unrolled
# scenes only (the other one is then taken from the regular build).
set -e
cd "$(dirname "$0")/../rust-ray-tracer_amd"
NAME=$1; shift
mkdir -p exp_libs
BASE="--offload-arch=gfx950 ${OPT:--O3} -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -fno-slp-vectorize"
if [ -z "$UNROLLED_ONLY" ]; then
    /opt/rocm/bin/hipcc $BASE -mllvm -amdgpu-sched-strategy=${SCHED:-iterative-ilp} "$@" -c csrc/rrt_kernels.hip -o exp_libs/$NAME.o
    K1=exp_libs/$NAME.o
else
    K1=csrc/rrt_kernels.o
fi
/opt/rocm/bin/hipcc $BASE "$@" -c csrc/rrt_kernels_unrolled.hip -o exp_libs/${NAME}_u.o
/opt/rocm/bin/hipcc $BASE -mllvm -amdgpu-sched-strategy=${UL_SCHED:-iterative-maxocc} "$@" -c csrc/rrt_kernels_unrolled_lists.hip -o exp_libs/${NAME}_ul.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o exp_libs/$NAME.so $K1 exp_libs/${NAME}_u.o exp_libs/${NAME}_ul.o csrc/lbvh_build.o csrc/rrt_api.o csrc/scene_build.o csrc/bvh_builder.o host/gltf_loader.o host/png_writer.o -lpthread -ldl
rm -f exp_libs/$NAME.o exp_libs/${NAME}_u.o exp_libs/${NAME}_ul.o
echo built exp_libs/$NAME.so
