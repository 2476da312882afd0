#!/bin/bash
# round 6: conv_wregd.hip with phases compiled out (scripts/exp_abl.sh conv_wregd DIR_WREGD_ABL 1 2 3 4 8 9 10)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6wregdabl}; mkdir -p $O
for b in "" 1 2 3 4 8 9 10; do
  if [ -z "$b" ]; then L=""; else L=$PWD/scripts/_exp/lib_conv_wregd_$b.so; fi
  [ -n "$b" ] && [ ! -f "$L" ] && continue
  EXP_SHAPES=l2.0 DIRTORCH_AMD_LIB=$L timeout 200 python scripts/exp_dual_time.py 2>&1 | grep -v "amdgpu.ids\|identical" | sed "s/^/DIR_WREGD_ABL=${b:-0}  /" >> $O/wregd_phases.txt
done
cat $O/wregd_phases.txt
