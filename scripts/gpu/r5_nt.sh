#!/bin/bash
# round 5: non-temporal cache policy (aux = 2) on the streaming loads / stores of the HBM-bound kernels, one kernel family at a time
# (scripts/exp_abl.sh <source> DIR_NT_LD 2 / DIR_NT_ST 2 -> scripts/_exp/lib_<source>_{ld,st}.so): A/B of the step + the affected layers
O=gpurun_out/${1:-r5nt}; mkdir -p $O
for rep in 1 2; do
for b in base conv_wreg_ld conv_wreg_st conv_c3c1_ld conv_c3c1_st conv_persist_ld; do
  if [ $b = base ]; then L=""; else L=$PWD/scripts/_exp/lib_$b.so; fi
  DIRTORCH_AMD_LIB=$L timeout 300 python bench.py --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --layers > $O/bench_${b}_$rep.json 2> $O/layers_${b}_$rep.txt
  echo "$b rep $rep: $(python -c "import json;d=json.loads(open('$O/bench_${b}_$rep.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])") $(grep -E 'layer3\.5\.conv3|layer2\.3\.conv3|layer1\.1\.c3c1|layer2\.1\.c3c1|layer3\.5\.conv1|layer4\.1\.conv3|layer3\.0\.ds' $O/layers_${b}_$rep.txt | awk '{print $3}' | tr '\n' ' ')"
done
done
