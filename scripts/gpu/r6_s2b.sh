#!/bin/bash
# round 6: conv_patchs2.hip's 64-channel-plane / 4-row form for 128 input channels (whole 128-byte lines per request) against its
# 32-channel-plane / 8-row form (DIRTORCH_AMD_PATCHS2_A) - parity, standalone, step A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6s2c}; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "patchs2 or 3x3_s2 or strided" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for m in B A; do
  if [ $m = A ]; then export DIRTORCH_AMD_PATCHS2_A=1; else unset DIRTORCH_AMD_PATCHS2_A; fi
  EXP_SHAPES=l2.0.conv2 timeout 300 python scripts/exp_conv_time.py 256x128_patchs2 256x128_w4x2_s3 2>&1 | grep conv2 | sed "s/^/form $m /"
done | tee $O/time.txt
for i in 1 2 3; do
  for m in B A; do
    if [ $m = A ]; then export DIRTORCH_AMD_PATCHS2_A=1; else unset DIRTORCH_AMD_PATCHS2_A; fi
    timeout 600 python bench.py --steps 30 --warmup 3 --cpu-seconds 0 --no-precision --layers > $O/bench_${m}_$i.json 2> $O/layers_${m}_$i.txt
  done
done
unset DIRTORCH_AMD_PATCHS2_A
grep -h "layer2.0.conv2" $O/layers_B_1.txt $O/layers_A_1.txt
python - <<P
import json
for m in ('B','A'):
    v=[]
    for i in (1,2,3):
        try: v.append(json.loads(open('$O/bench_%s_%d.json'%(m,i)).read().strip().splitlines()[-1])['value'])
        except Exception as e: v.append(str(e)[:60])
    print(m, v)
P
