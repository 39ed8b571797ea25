#!/bin/bash
# Runs bench.py for the given "config[:spp]" items and prints a one-line digest each (extra environment: as given).
#   bash tools/bench_all.sh 1 2 3 4 4:128 5:16 6:16
cd "$GRAFT_REPO_ROOT" || exit 1
for item in "$@"; do
  c=${item%%:*}; spp=""; [ "$item" != "$c" ] && spp="--spp ${item##*:}"
  timeout -k 10 400 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --config $c $spp > gpurun_out/bench_c$c.json 2> gpurun_out/bench_c$c.err || { echo FAIL $item; tail -5 gpurun_out/bench_c$c.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/bench_c$c.json')); r=d['roofline']
print('config', '$item', round(d['value']), 'Mrays/s', round(d['ms_per_step'],2), 'ms/step kernel', round(r['kernel_ms'],2), 'rays/sample', round(d['rays_per_sample'],3), 'nodes/ray', round(r['nodes_per_ray'],1), 'prims/ray', round(r['prims_per_ray'],2), 'bound', r.get('bound'), 'frac', r.get('frac') and round(r['frac'],3), 'build_ms', round(d['config']['bvh_build_ms']))"
done
