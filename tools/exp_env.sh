#!/bin/bash
# Sweeps one environment variable over values for a bench config (experiment helper).
#   CFG=4 SPP=128 bash tools/exp_env.sh RRT_LEAVE_WALK 8 12 16
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; shift
SPPARG=""; [ -n "$SPP" ] && SPPARG="--spp $SPP"
for v in "$@"; do
  env $VAR=$v timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --config ${CFG:-2} $SPPARG > gpurun_out/exp_env.json 2> gpurun_out/exp_env.err || { echo FAIL $v; tail -3 gpurun_out/exp_env.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/exp_env.json')); r=d['roofline']
print('cfg', ${CFG:-2}, '$VAR', '$v', 'kernel_ms', round(r['kernel_ms'],2), 'Mrays/s', round(d['value']), 'util node/prim/shade', round(r['lane_util_node'],3), round(r['lane_util_prim'],3), round(r['lane_util_shade'],3), 'wave steps/ray', {k: round(v,1) for k,v in r['wave_steps_per_ray'].items()})"
done
