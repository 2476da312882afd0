#!/bin/bash
# round 6: is the 2 KB-stride read pattern of the wide 1x1 convs (128-byte pieces of 256 pixel rows per K-stage) what keeps
# layer3.conv1 at 3.8 TB/s?  conv_persist.hip with its pixel operand addressed as [M/256][Cin/64][256][64] (every K-stage one
# contiguous 32 KB; timing only - scripts/exp_abl.sh conv_persist DIR_PERSIST_BLOCKED 1) against the default build.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6blocked}; mkdir -p $O
for L in "" $PWD/scripts/_exp/lib_conv_persist_1.so; do
  echo "== lib ${L:-default}" | tee -a $O/blocked.txt
  EXP_SHAPES=l3.conv1,l4.conv1,l3.0.conv1 DIRTORCH_AMD_LIB=$L timeout 300 python scripts/exp_conv_time.py 256x256_persist1x1 256x256_persist1x1_x3 2>&1 | grep -v amdgpu.ids | tee -a $O/blocked.txt
done
