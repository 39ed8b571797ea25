#!/bin/bash
# Kernel trace of one quick.py run: bash tools/r3_kt.sh <label> <quick.py args...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
L=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$L -- python3 tools/quick.py "$@" > gpurun_out/kt_$L.log 2>&1 || { tail -5 gpurun_out/kt_$L.log; exit 1; }
python3 - gpurun_out/kt_$L <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg_ms {float(r['AverageNs'])/1e6:10.3f}")
PY
