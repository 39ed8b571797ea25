#!/bin/bash
# Times one bench configuration under several values of one environment knob.
#   bash tools/r2_env.sh <label> <config> <spp> <ENVNAME> v1 v2 ...
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="gpurun_out/$1"; CFG=$2; SPP=$3; KNOB=$4; shift 4
mkdir -p "$OUT"
for v in "$@"; do
    env $KNOB=$v timeout -k 10 300 python3 bench.py --config $CFG --spp $SPP --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/c${CFG}_${KNOB}_$v.json" 2> "$OUT/c${CFG}_${KNOB}_$v.err" || { echo "$v failed"; tail -5 "$OUT/c${CFG}_${KNOB}_$v.err"; exit 1; }
    python3 -c "
import json
d=json.loads(open('$OUT/c${CFG}_${KNOB}_$v.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$KNOB=$v', 'kernel_ms %.2f' % r['kernel_ms'], 'step_ms %.2f' % d['ms_per_step'], 'Grays/s %.3f' % (d['value']/1e3))"
done
