#!/bin/bash
# Runs bench.py for the given configs and prints a one-line digest each.
cd "$GRAFT_REPO_ROOT" || exit 1
for c in "$@"; do
  timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --config $c > gpurun_out/bench_c$c.json 2> gpurun_out/bench_c$c.err || { echo FAIL $c; tail -5 gpurun_out/bench_c$c.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_c$c.json')); r=d['roofline']
print('config', $c, round(d['value']), 'Mrays/s', round(d['ms_per_step'],2), 'ms/step kernel', round(r['kernel_ms'],2), 'rays/sample', round(d['rays_per_sample'],3), 'nodes/ray', round(r['nodes_per_ray'],1), 'prims/ray', round(r['prims_per_ray'],2), 'algGB/s', round(r['achieved']), 'build_ms', round(d['config']['bvh_build_ms']))"
done
