#!/bin/bash
# Sweep of one environment knob over bench configurations: bash tools/r3_sweep.sh <label> <VAR> "<values>" "<configs>"
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="gpurun_out/$1"; VAR=$2; VALUES=$3; CONFIGS=$4
mkdir -p "$OUT"
for cfg in $CONFIGS; do
    spp=128; [ "$cfg" = 5 ] && spp=16; [ "$cfg" = 6 ] && spp=16; [ "$cfg" = 2 ] && spp=256
    for v in $VALUES; do
        env $VAR=$v timeout -k 10 300 python3 bench.py --config $cfg --spp $spp --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/c${cfg}_$v.json" 2> "$OUT/c${cfg}_$v.err" || { echo "bench $cfg $v failed"; tail -5 "$OUT/c${cfg}_$v.err"; exit 1; }
        python3 - "$OUT/c${cfg}_$v.json" "$VAR=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{sys.argv[2]:24s} config {d['config']['workload'][:28]:28s} kernel_ms {r['kernel_ms']:8.2f}  Grays/s {d['value']/1e3:6.2f}  nodes/prims per ray {r['nodes_per_ray']:.1f}/{r['prims_per_ray']:.2f} util n/p/s {r['lane_util_node']:.2f}/{r['lane_util_prim']:.2f}/{r['lane_util_shade']:.2f}")
PY
    done
done
