#!/bin/bash
# round 5: MFMA issue order in the paired stem (scripts/exp_abl.sh conv_pair DIR_STEMP_ORDER 1 2), standalone
O=gpurun_out/${1:-r5stemorder}; mkdir -p $O
for rep in 1 2; do
for b in "" 1 2; do
  if [ -z "$b" ]; then L=""; else L=$PWD/scripts/_exp/lib_conv_pair_$b.so; fi
  DIRTORCH_AMD_LIB=$L timeout 200 python scripts/exp_stem_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/order ${b:-0} rep $rep  /" | tee -a $O/stem_order.txt
done
done
