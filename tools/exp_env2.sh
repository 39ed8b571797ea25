#!/bin/bash
# Sweeps "VAR=value[,VAR2=value2...]" settings over a bench config (experiment helper).
#   CFG=4 bash tools/exp_env2.sh RRT_LEAF_TRIS=2,RRT_SAH_TCOST=0.5 RRT_LEAF_TRIS=4
cd "$GRAFT_REPO_ROOT" || exit 1
for setting in "$@"; do
  env ${setting//,/ } timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --config ${CFG:-2} > gpurun_out/exp_env.json 2> gpurun_out/exp_env.err || { echo FAIL $setting; tail -3 gpurun_out/exp_env.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/exp_env.json')); r=d['roofline']
print('cfg', ${CFG:-2}, '$setting', 'kernel_ms', round(r['kernel_ms'],2), 'Mrays/s', round(d['value']), 'nodes/prims per ray', round(r['nodes_per_ray'],1), round(r['prims_per_ray'],2), 'util', round(r['lane_util_node'],3), round(r['lane_util_prim'],3), 'depth', d['config']['bvh_depth'], 'nodes', d['config']['bvh_nodes'])"
done
