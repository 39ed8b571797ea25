#!/bin/bash
# Round-3 profiles of every bench configuration: bash tools/r3_prof_all.sh  (run under gpurun; output in gpurun_out/r3prof_*)
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/prof.sh r3prof_c4 --config 4 > gpurun_out/r3prof_c4.txt 2>&1 || { tail -5 gpurun_out/r3prof_c4.txt; exit 1; }
echo "c4 done"
bash tools/prof.sh r3prof_c4_128 --config 4 --spp 128 > gpurun_out/r3prof_c4_128.txt 2>&1 || { tail -5 gpurun_out/r3prof_c4_128.txt; exit 1; }
echo "c4@128 done"
bash tools/prof.sh r3prof_c2 --config 2 > gpurun_out/r3prof_c2.txt 2>&1 || { tail -5 gpurun_out/r3prof_c2.txt; exit 1; }
echo "c2 done"
bash tools/prof.sh r3prof_c3 --config 3 > gpurun_out/r3prof_c3.txt 2>&1 || { tail -5 gpurun_out/r3prof_c3.txt; exit 1; }
echo "c3 done"
bash tools/prof.sh r3prof_c6 --config 6 --spp 16 > gpurun_out/r3prof_c6.txt 2>&1 || { tail -5 gpurun_out/r3prof_c6.txt; exit 1; }
echo "c6 done"
bash tools/prof.sh r3prof_c5 --config 5 --spp 16 > gpurun_out/r3prof_c5.txt 2>&1 || { tail -5 gpurun_out/r3prof_c5.txt; exit 1; }
echo "c5 done"
