#!/bin/bash
# Profiles of the configurations whose kernel takes per-pixel candidate lists: bash tools/r3_prof_lists.sh  (under gpurun)
# (RRT_LIST_WAVES fixes the build of the kernel the library settles on for the scene, so that every pass profiles it.)
cd "$GRAFT_REPO_ROOT" || exit 1
export RRT_LIST_WAVES=7
bash tools/prof.sh r3lprof_c4 --config 4 > gpurun_out/r3lprof_c4.txt 2>&1 || { tail -5 gpurun_out/r3lprof_c4.txt; exit 1; }
echo "c4 done"
bash tools/prof.sh r3lprof_c4_128 --config 4 --spp 128 > gpurun_out/r3lprof_c4_128.txt 2>&1 || { tail -5 gpurun_out/r3lprof_c4_128.txt; exit 1; }
echo "c4@128 done"
export RRT_LIST_WAVES=7
bash tools/prof.sh r3lprof_c3 --config 3 > gpurun_out/r3lprof_c3.txt 2>&1 || { tail -5 gpurun_out/r3lprof_c3.txt; exit 1; }
echo "c3 done"
unset RRT_LIST_WAVES
bash tools/prof.sh r3lprof_c2 --config 2 > gpurun_out/r3lprof_c2.txt 2>&1 || { tail -5 gpurun_out/r3lprof_c2.txt; exit 1; }
echo "c2 done"
