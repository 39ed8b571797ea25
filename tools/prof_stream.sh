#!/bin/bash
# PMC passes over the ray-batch kernel on tile-ordered primary rays (tools/exp_stream_prof.py); one rocprofv3 run per pass.
#   bash tools/prof_stream.sh <out_dir_under_gpurun_out>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT="gpurun_out/$1"; shift
mkdir -p "$OUT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 tools/exp_stream_prof.py 4 16777216 3 > "$OUT/kt.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$OUT/kt.log"; exit 1; }
pass() {
    name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 tools/exp_stream_prof.py 4 16777216 1 > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass pmc1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pass pmc2 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
pass pmc3 TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
RRT_PROF_KERNEL=k_trace_stream python3 tools/prof_summary.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
