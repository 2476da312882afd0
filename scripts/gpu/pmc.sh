#!/bin/bash
# SQ / cache counters per kernel for bench.py (diagnosis; separate --pmc passes, kernel-trace only)
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
export DIRTORCH_AMD_TUNE_CACHE=$R/gpurun_out/tune_pmc.txt
ARGS="--steps 2 --warmup 1 --cpu-seconds 0 --autotune"
timeout 300 python bench.py $ARGS > /dev/null 2>&1   # writes the tuning cache
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_sq -o bench -- python bench.py $ARGS > /dev/null 2> gpurun_out/pmc_sq.err
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum -d $R/gpurun_out/pmc_tc -o bench -- python bench.py $ARGS > /dev/null 2> gpurun_out/pmc_tc.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_sq2 -o bench -- python bench.py $ARGS > /dev/null 2> gpurun_out/pmc_sq2.err
tail -3 gpurun_out/pmc_sq.err gpurun_out/pmc_tc.err gpurun_out/pmc_sq2.err
ls -la gpurun_out/pmc_*
