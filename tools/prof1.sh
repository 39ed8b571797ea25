cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof1/kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof1/kt.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/prof1/pmc1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof1/pmc1.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof1/pmc2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof1/pmc2.log 2>&1 && \
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d gpurun_out/prof1/pmc3 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof1/pmc3.log 2>&1
echo rc=$?
find gpurun_out/prof1 -name "*.csv" | head -20
