#!/bin/bash
# Times library variants (rust-ray-tracer_amd/exp_libs/<name>.so, "base" = the in-tree build) on bench configurations:
#   bash tools/r3_variants.sh "<variants>" "<config:spp[,spp] ...>" [quick.py options]
cd "$GRAFT_REPO_ROOT" || exit 1
VARIANTS=$1; CASES=$2; shift 2
for case in $CASES; do
    cfg=${case%%:*}; spp=${case##*:}
    for v in $VARIANTS; do
        if [ "$v" = base ]; then unset RRT_LIB_PATH; else export RRT_LIB_PATH="$GRAFT_REPO_ROOT/rust-ray-tracer_amd/exp_libs/$v.so"; fi
        echo "== variant $v"
        timeout -k 10 400 python3 -u tools/quick.py --config $cfg --spp $spp "$@" || { echo "FAILED $v $case"; exit 1; }
    done
done
