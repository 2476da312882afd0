#!/bin/bash
# round 6: the uint8 stem with phases compiled out (scripts/exp_abl.sh stem_u8 DIR_STEMU8_ABL 1 2 3 4 6 7) and its segment lengths
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6stemabl}; mkdir -p $O
for b in "" 1 2 3 4 6 7; do
  if [ -z "$b" ]; then L=""; else L=$PWD/scripts/_exp/lib_stem_u8_$b.so; fi
  [ -n "$b" ] && [ ! -f "$L" ] && continue
  DIRTORCH_AMD_LIB=$L timeout 200 python scripts/exp_stem_u8_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/DIR_STEMU8_ABL=${b:-0}  /" >> $O/stem_u8_phases.txt
done
for seg in 1 2 4 16; do
  DIRTORCH_AMD_STEM_U8_SEG=$seg timeout 200 python scripts/exp_stem_u8_time.py 2>&1 | grep -v amdgpu.ids >> $O/stem_u8_phases.txt
done
DIRTORCH_AMD_STEM_U8_WG8=1 timeout 200 python scripts/exp_stem_u8_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/one 8-wave workgroup per CU  /" >> $O/stem_u8_phases.txt
EXP_PICS=1 timeout 200 python scripts/exp_stem_u8_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/synthetic pictures  /" >> $O/stem_u8_phases.txt
DIRTORCH_AMD_NO_STEM_U8=1 timeout 200 python scripts/exp_stem_u8_time.py 2>&1 | grep -v amdgpu.ids | sed "s/^/generic paired stem  /" >> $O/stem_u8_phases.txt
cat $O/stem_u8_phases.txt
