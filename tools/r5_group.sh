#!/bin/bash
# GPU builder: size of the groups of the second round (RRT_LBVH_GROUP; 0: the host arranges all subtrees).
#   bash tools/r5_group.sh <label> <config> <group sizes...>
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="gpurun_out/$1"; cfg=$2; shift 2
mkdir -p "$OUT"
spp=128; [ "$cfg" = 5 ] && spp=16; [ "$cfg" = 6 ] && spp=16
for gsz in "$@"; do
    RRT_LBVH_GROUP=$gsz timeout -k 10 300 python3 bench.py --config $cfg --spp $spp --steps 3 --warmup 1 --no-cpu-baseline --device-build > "$OUT/c${cfg}_g$gsz.json" 2> "$OUT/c${cfg}_g$gsz.err" || { echo "bench failed"; tail -n 5 "$OUT/c${cfg}_g$gsz.err"; exit 1; }
    python3 -c "
import json
d=json.loads(open('$OUT/c${cfg}_g$gsz.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('config $cfg group %-3s' % '$gsz', 'kernel_ms %8.2f' % r['kernel_ms'], 'build_ms %7.1f' % c['bvh_build_ms'], 'depth', c['bvh_depth'], 'nodes/ray %.1f prims/ray %.2f' % (r['nodes_per_ray'], r['prims_per_ray']))"
done
