#!/bin/bash
# round 6: the stride-2 3x3 patch kernel (conv_patchs2.hip) - parity, standalone timing against the generic tiles, step A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6s2}; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "patchs2 or 3x3_s2" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
EXP_SHAPES=l2.0.conv2,l3.0.conv2,l4.0.conv2 timeout 300 python scripts/exp_conv_time.py 256x128_patchs2 256x128_w4x2_s3 256x256_w4x4 128x128_w2x2 > $O/time.txt 2>&1; cat $O/time.txt
for i in 1 2; do
  for m in new old; do
    if [ $m = old ]; then export DIRTORCH_AMD_NO_PATCHS2=1; else unset DIRTORCH_AMD_NO_PATCHS2; fi
    timeout 600 python bench.py --steps 30 --warmup 3 --cpu-seconds 0 --no-precision > $O/bench_${m}_$i.json 2> $O/err_${m}_$i.txt
  done
done
unset DIRTORCH_AMD_NO_PATCHS2
python - <<P
import json
for m in ('new','old'):
    v=[]
    for i in (1,2):
        try: v.append(json.loads(open('$O/bench_%s_%d.json'%(m,i)).read().strip().splitlines()[-1])['value'])
        except Exception as e: v.append(str(e)[:60])
    print(m, v)
P
