#!/bin/bash
# round 5: wave priorities in the loader / consumer form of conv_patch3x3w (scripts/exp_abl.sh conv_patchw DIR_PATCHW_PRIO 1 2 3): A/B of the step
O=gpurun_out/${1:-r5prio}; mkdir -p $O
for rep in 1 2; do
for b in "" 1 2 3; do
  if [ -z "$b" ]; then L=""; else L=$PWD/scripts/_exp/lib_conv_patchw_$b.so; fi
  DIRTORCH_AMD_LIB=$L timeout 300 python bench.py --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --layers > $O/bench_p${b:-0}_$rep.json 2> $O/layers_p${b:-0}_$rep.txt
  echo "prio ${b:-0} rep $rep: $(python -c "import json;d=json.loads(open('$O/bench_p${b:-0}_$rep.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])") $(grep -E 'layer3\.5\.conv2' $O/layers_p${b:-0}_$rep.txt | awk '{print $3}')"
done
done
