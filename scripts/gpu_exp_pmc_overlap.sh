#!/bin/bash
# PMC: why do fill and MFMA phases of the 3x3 conv not overlap?  full vs fill-only vs compute-only builds
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
V="256x256_w4x2 256x256_w4x4"
for mode in full FILL_ONLY NO_FILL; do
  if [ $mode = full ]; then unset DIRTORCH_AMD_LIB; else export DIRTORCH_AMD_LIB=$R/scripts/_exp/libdir_$mode.so; fi
  rm -rf gpurun_out/pmcx_$mode
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmcx_$mode -o x -- python scripts/exp_conv_time.py $V > /dev/null 2> gpurun_out/pmcx_$mode.err
  rm -rf gpurun_out/pmcy_$mode
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $R/gpurun_out/pmcy_$mode -o y -- python scripts/exp_conv_time.py $V > /dev/null 2> gpurun_out/pmcy_$mode.err
  tail -2 gpurun_out/pmcx_$mode.err gpurun_out/pmcy_$mode.err
done
for mode in full FILL_ONLY NO_FILL; do echo "== $mode"; python scripts/pmc_table.py gpurun_out/pmcx_$mode gpurun_out/pmcy_$mode 2>&1 | grep -i "igemm\|kernel\|^-" | head -12; done | tee gpurun_out/pmc_overlap.txt
