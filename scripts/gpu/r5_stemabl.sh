#!/bin/bash
# round 5: the paired stem with phases compiled out (scripts/exp_abl.sh conv_pair DIR_STEMP_ABL 1 2 3 4)
O=gpurun_out/${1:-r5stem}; mkdir -p $O
for b in "" 1 2 3 4; do
  if [ -z "$b" ]; then L=""; else L=$PWD/scripts/_exp/lib_conv_pair_$b.so; fi
  DIRTORCH_AMD_LIB=$L timeout 200 python scripts/exp_stem_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/DIR_STEMP_ABL=${b:-0}  /" >> $O/stem_phases.txt
done
cat $O/stem_phases.txt
