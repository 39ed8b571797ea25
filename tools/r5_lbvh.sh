#!/bin/bash
# GPU builder with the subtrees below the cut rebuilt on the GPU: parity tests with verified trees, then render and
# build time against the radix topology below a cut of 16 (the former default).
#   bash tools/r5_lbvh.sh <label> <configs...>
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="gpurun_out/$1"; shift
mkdir -p "$OUT"
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -u -m pytest tests/test_gpu_device_build.py -x -q --timeout 400 --timeout-method thread > "$OUT/tests.log" 2>&1 || { echo "tests failed"; tail -n 40 "$OUT/tests.log"; exit 1; }
tail -n 1 "$OUT/tests.log"
fi
for cfg in "$@"; do
  spp=128; [ "$cfg" = 5 ] && spp=16; [ "$cfg" = 6 ] && spp=16
  for variant in ${VARIANTS:-"sah64" "radix16"}; do
    case $variant in
      sah*) export RRT_LBVH_CLUSTER=${variant#sah}; export RRT_LBVH_SAH=1;;
      radix*) export RRT_LBVH_CLUSTER=${variant#radix}; export RRT_LBVH_SAH=0;;
    esac
    RRT_LBVH_VERIFY=${VERIFY:-0} timeout -k 10 300 python3 bench.py --config $cfg --spp $spp --steps 3 --warmup 1 --no-cpu-baseline --device-build > "$OUT/c${cfg}_$variant.json" 2> "$OUT/c${cfg}_$variant.err" || { echo "bench failed"; tail -n 5 "$OUT/c${cfg}_$variant.err"; exit 1; }
    python3 -c "
import json
d=json.loads(open('$OUT/c${cfg}_$variant.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('config $cfg %-8s' % '$variant', 'kernel_ms %8.2f' % r['kernel_ms'], 'Grays/s %6.2f' % (d['value']/1e3), 'build_ms %7.1f' % c['bvh_build_ms'], 'depth', c['bvh_depth'], 'nodes/ray %.1f prims/ray %.2f' % (r['nodes_per_ray'], r['prims_per_ray']))"
  done
done
