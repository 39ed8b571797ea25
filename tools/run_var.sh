#!/bin/bash
# Times bench configuration $1 at $2 spp with every library under rust-ray-tracer_amd/exp_libs (and the regular one).
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "" rust-ray-tracer_amd/exp_libs/*.so; do
  echo "== ${lib:-regular}" >> gpurun_out/var.log
  RRT_LIB_PATH=$lib timeout -k 10 200 python -u tools/quick.py --config $1 --spp $2 --reps 2 >> gpurun_out/var.log 2>&1 || exit 1
done
