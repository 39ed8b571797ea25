#!/bin/bash
# Profiles one bench configuration with rocprofv3: kernel trace + stats, then PMC passes (one run each).
#   bash tools/prof.sh <out_dir_under_gpurun_out> <bench args...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="gpurun_out/$1"; shift
ARGS="$*"
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-frame-check $ARGS > "$OUT/kt.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$OUT/kt.log"; exit 1; }
pass() {
    name=$1; shift
    timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-frame-check $ARGS > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass pmc1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pass pmc2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
pass pmc3 TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass pmc4 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass pmc5 SQ_INSTS_VALU SQ_WAIT_INST_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_FLAT
# HBM-side traffic: FETCH_SIZE takes 3 TCC slots and WRITE_SIZE 2, so one pass each (values in KB)
pass pmc6 FETCH_SIZE
pass pmc7 WRITE_SIZE
# the address / data path of the vector L1 (one TA and one TD per CU): busy cycles, summed over the units and averaged
pass pmc8 TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
