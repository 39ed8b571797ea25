#!/bin/bash
# Times one bench configuration with several builds of the library (kernel experiments).
#   bash tools/r2_libs.sh <label> <config> <spp> lib1.so lib2.so ...
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="gpurun_out/$1"; CFG=$2; SPP=$3; shift 3
mkdir -p "$OUT"
for lib in "$@"; do
    name=$(basename "$lib" .so)
    RRT_LIB_PATH="$PWD/rust-ray-tracer_amd/$lib" timeout -k 10 300 python3 bench.py --config $CFG --spp $SPP --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/c${CFG}_$name.json" 2> "$OUT/c${CFG}_$name.err" || { echo "$name failed"; tail -5 "$OUT/c${CFG}_$name.err"; exit 1; }
    python3 -c "
import json,sys
d=json.loads(open('$OUT/c${CFG}_$name.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$name', 'kernel_ms %.2f' % r['kernel_ms'], 'Grays/s %.2f' % (d['value']/1e3), 'util %.2f/%.2f/%.2f' % (r['lane_util_node'], r['lane_util_prim'], r['lane_util_shade']))"
done
