#!/bin/bash
# Quick regression + timing loop for kernel changes: the render / intersect parity tests, then config 4 at 128 spp.
#   bash tools/r2_quick.sh <label> [extra bench configs...]
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="gpurun_out/$1"; shift
mkdir -p "$OUT"
timeout -k 10 600 python -u -m pytest tests/test_gpu_render.py tests/test_gpu_intersect.py tests/test_gpu_device_build.py -x -q --timeout 300 --timeout-method thread > "$OUT/tests.log" 2>&1 || { echo "tests failed"; tail -30 "$OUT/tests.log"; exit 1; }
tail -1 "$OUT/tests.log"
for cfg in 4 "$@"; do
    spp=128; [ "$cfg" = 5 ] && spp=16; [ "$cfg" = 6 ] && spp=16; [ "$cfg" = 2 ] && spp=256
    timeout -k 10 300 python3 bench.py --config $cfg --spp $spp --steps 5 --warmup 1 --no-cpu-baseline > "$OUT/bench_c$cfg.json" 2> "$OUT/bench_c$cfg.err" || { echo "bench $cfg failed"; tail -5 "$OUT/bench_c$cfg.err"; exit 1; }
    python3 - "$OUT/bench_c$cfg.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"config {d['config']['workload'][:40]:40s} kernel_ms {r['kernel_ms']:8.2f}  Grays/s {d['value']/1e3:6.2f}  nodes/ray {r['nodes_per_ray']:.1f} util n/p/s {r['lane_util_node']:.2f}/{r['lane_util_prim']:.2f}/{r['lane_util_shade']:.2f}")
PY
done
