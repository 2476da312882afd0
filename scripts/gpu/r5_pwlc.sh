#!/bin/bash
# round 5: the loader / consumer form of conv_patch3x3w - parity (bit-identical to the one-role kernel), standalone timing, A/B of the step
O=gpurun_out/${1:-r5pwlc}; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "patchw or patch3x3w or full_size" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in "" 1; do
  echo "== DIRTORCH_AMD_NO_PATCHW_LC=$v" >> $O/time.txt
  DIRTORCH_AMD_NO_PATCHW_LC=$v EXP_SHAPES=l3.conv2,l2.conv2,l4.conv2 timeout 200 python scripts/exp_conv_time.py 512x128_patch3x3w 2>&1 | grep -v "amdgpu.ids" >> $O/time.txt
done
cat $O/time.txt
for rep in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --layers > $O/bench_lc_$rep.json 2> $O/layers_lc_$rep.txt
  DIRTORCH_AMD_NO_PATCHW_LC=1 timeout 300 python bench.py --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --layers > $O/bench_one_$rep.json 2> $O/layers_one_$rep.txt
done
for f in $O/bench_*.json; do echo $f $(python -c "import json;d=json.loads(open('$f').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"); done
grep -E "layer3\.(5|6)\.conv2|layer2\.2\.conv2|layer4\.1\.conv2" $O/layers_lc_1.txt $O/layers_one_1.txt
