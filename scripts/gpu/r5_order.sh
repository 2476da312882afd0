#!/bin/bash
# round 5: MFMA issue order in conv_patch3x3w_lc (scripts/exp_abl.sh conv_patchw DIR_PW_ORDER 1: the pixel fragment held over four
# consecutive MFMAs) - the kernel is power-bound, does the operand that toggles matter?  A/B of the step, interleaved.
O=gpurun_out/${1:-r5order}; mkdir -p $O
for rep in 1 2; do
for b in "" 1 2 3; do
  if [ -z "$b" ]; then L=""; else L=$PWD/scripts/_exp/lib_conv_patchw_$b.so; fi
  DIRTORCH_AMD_LIB=$L timeout 300 python bench.py --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --layers > $O/bench_o${b:-0}_$rep.json 2> $O/layers_o${b:-0}_$rep.txt
  echo "order ${b:-0} rep $rep: $(python -c "import json;d=json.loads(open('$O/bench_o${b:-0}_$rep.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])") $(grep -E 'layer3\.5\.conv2|layer4\.1\.conv2' $O/layers_o${b:-0}_$rep.txt | awk '{print $3}' | tr '\n' ' ')"
done
done
