#!/bin/bash
# GPU builder: leaf size x cluster size on one configuration.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="gpurun_out/$1"; cfg=$2; shift 2
mkdir -p "$OUT"
spp=128; [ "$cfg" = 5 ] && spp=16; [ "$cfg" = 6 ] && spp=16
for leaf in ${LEAVES:-1 2 3 4}; do for cl in "$@"; do
    RRT_LBVH_LEAF=$leaf RRT_LBVH_CLUSTER=$cl timeout -k 10 300 python3 bench.py --config $cfg --spp $spp --steps 3 --warmup 1 --no-cpu-baseline --device-build > "$OUT/c${cfg}_l${leaf}_c$cl.json" 2> "$OUT/c${cfg}_l${leaf}_c$cl.err" || { echo "bench failed"; tail -5 "$OUT/c${cfg}_l${leaf}_c$cl.err"; exit 1; }
    python3 -c "
import json
d=json.loads(open('$OUT/c${cfg}_l${leaf}_c$cl.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('config $cfg leaf $leaf cluster %-4s' % '$cl', 'kernel_ms %8.2f' % r['kernel_ms'], 'Grays/s %6.2f' % (d['value']/1e3), 'build_ms %7.1f' % c['bvh_build_ms'], 'depth', c['bvh_depth'], 'nodes/ray %.1f prims/ray %.2f' % (r['nodes_per_ray'], r['prims_per_ray']))"
done; done
