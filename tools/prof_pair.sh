#!/bin/bash
# One PMC pass (load / LDS / vector instruction counts, L1 address-unit busy cycles) of a bench configuration under two
# values of an environment variable: what a knob does to the counters, not just to the time.
#   bash tools/prof_pair.sh <out_dir_under_gpurun_out> <VAR> <value_a> <value_b> <bench args...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="gpurun_out/$1"; VAR=$2; A=$3; B=$4; shift 4
mkdir -p "$OUT"
for v in "$A" "$B"; do
    export $VAR=$v
    timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d "$OUT/v_$v" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > "$OUT/v_$v.log" 2>&1 || { echo "pass $v failed"; tail -5 "$OUT/v_$v.log"; exit 1; }
    python3 - "$OUT/v_$v" "$VAR=$v" <<'PY'
import csv, glob, sys
rows = {}
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_render" in n and ", false," in n.split("k_render")[1][:12]:
            rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
d = rows[sorted(rows, key=int)[0]]
cyc = d["GRBM_GUI_ACTIVE"] / 8
print(sys.argv[2], {k: f"{v:.4g}" for k, v in d.items()}, "ta_busy_frac %.3f" % (d["TA_BUSY_avr"] / cyc), "l1_floor %.3f" % (d["SQ_INSTS_VMEM_RD"] * 16 / (256 * cyc)), "kernel_ms@2.4GHz %.1f" % (cyc / 2.4e6))
PY
done
