#!/bin/bash
# round 5: conv_patch3x3w with phases compiled out (64 = no LDS-DMA, 128 = no fragment reads / MFMAs, 66 = 64 + no epilogue)
O=gpurun_out/${1:-r5pw}; mkdir -p $O
for b in "" 64 128 66; do
  if [ -z "$b" ]; then L=""; else L=$PWD/scripts/_exp/lib_conv_patchw_$b.so; fi
  echo "== DIR_PATCHW_ABL=${b:-0}" >> $O/patchw_phases.txt
  DIRTORCH_AMD_LIB=$L EXP_SHAPES=l3.conv2,l2.conv2,l4.conv2 timeout 200 python scripts/exp_conv_time.py 512x128_patch3x3w 2>&1 | grep -v "amdgpu.ids" >> $O/patchw_phases.txt
done
cat $O/patchw_phases.txt
