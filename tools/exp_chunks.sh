#!/bin/bash
# Sweeps the number of sample chunks of the render kernel (experiment helper).
cd "$GRAFT_REPO_ROOT" || exit 1
for c in "$@"; do
  RRT_CHUNKS=$c timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --config ${CFG:-2} > gpurun_out/exp_$c.json 2> gpurun_out/exp_$c.err || { echo FAIL $c; tail -3 gpurun_out/exp_$c.err; exit 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/exp_$c.json')); print('chunks', $c, 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'Mrays/s', round(d['value']))"
done
