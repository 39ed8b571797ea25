#!/bin/bash
# Host SAH build against the GPU linear-BVH build: build time and render rate for the given configs.
cd "$GRAFT_REPO_ROOT" || exit 1
for c in "$@"; do
  for mode in "" "--device-build"; do
    timeout -k 10 400 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --config $c $mode > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err || { echo FAIL $c $mode; tail -5 gpurun_out/bench_b.err; exit 1; }
    python3 -c "
import json; d=json.load(open('gpurun_out/bench_b.json')); r=d['roofline']; c=d['config']
print('config', $c, c['bvh_builder'], 'build_ms', round(c['bvh_build_ms'],1), 'depth', c['bvh_depth'], 'nodes', c['bvh_nodes'], 'kernel_ms', round(r['kernel_ms'],2), 'Mrays/s', round(d['value']), 'nodes/prims per ray', round(r['nodes_per_ray'],1), round(r['prims_per_ray'],2))"
  done
done
