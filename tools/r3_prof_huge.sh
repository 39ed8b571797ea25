#!/bin/bash
# Profiles of the ten-million-triangle configurations at 256 spp, where their primary rays take per-pixel candidate
# lists: bash tools/r3_prof_huge.sh  (under gpurun)
cd "$GRAFT_REPO_ROOT" || exit 1
export RRT_LIST_WAVES=7
bash tools/prof.sh r3hprof_c6 --config 6 --spp 256 > gpurun_out/r3hprof_c6.txt 2>&1 || { tail -5 gpurun_out/r3hprof_c6.txt; exit 1; }
echo "c6 done"
bash tools/prof.sh r3hprof_c5 --config 5 --spp 256 > gpurun_out/r3hprof_c5.txt 2>&1 || { tail -5 gpurun_out/r3hprof_c5.txt; exit 1; }
echo "c5 done"
