#!/bin/bash
# GPU builder: parity tests, then render speed and build time per cluster size against the host SAH build.
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="gpurun_out/$1"; shift
mkdir -p "$OUT"
timeout -k 10 600 python -u -m pytest tests/test_gpu_device_build.py -x -q --timeout 300 --timeout-method thread > "$OUT/tests.log" 2>&1 || { echo "tests failed"; tail -30 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
for cfg in "$@"; do
    spp=128; [ "$cfg" = 5 ] && spp=16; [ "$cfg" = 6 ] && spp=16
    for mode in host 0 16 64 256; do
        flag="--device-build"; [ "$mode" = host ] && flag=""
        RRT_LBVH_CLUSTER=$mode timeout -k 10 300 python3 bench.py --config $cfg --spp $spp --steps 3 --warmup 1 --no-cpu-baseline $flag > "$OUT/c${cfg}_$mode.json" 2> "$OUT/c${cfg}_$mode.err" || { echo "bench $cfg $mode failed"; tail -5 "$OUT/c${cfg}_$mode.err"; exit 1; }
        python3 -c "
import json
d=json.loads(open('$OUT/c${cfg}_$mode.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('config $cfg builder %-5s' % '$mode', 'kernel_ms %8.2f' % r['kernel_ms'], 'Grays/s %6.2f' % (d['value']/1e3), 'build_ms %7.1f' % c['bvh_build_ms'], 'depth', c['bvh_depth'], 'nodes/ray %.1f prims/ray %.2f' % (r['nodes_per_ray'], r['prims_per_ray']))"
    done
done
