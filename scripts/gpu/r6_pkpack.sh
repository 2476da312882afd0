#!/bin/bash
# round 6: conv_persist.hip's weight stages from a packed copy (contiguous KBs per DMA instruction) - step A/B with the golden tests as parity.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6pkpack}; mkdir -p $O
DIRTORCH_AMD_PERSIST_PACK=1 DIRTORCH_AMD_PATCHW_PACK=1 timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for m in plain packed; do
  if [ $m = packed ]; then export DIRTORCH_AMD_PERSIST_PACK=1; else unset DIRTORCH_AMD_PERSIST_PACK; fi
  EXP_SHAPES=l3.conv1,l4.conv1,l3.0.conv1 timeout 300 python scripts/exp_conv_time.py 256x256_persist1x1 256x256_persist1x1_x3 2>&1 | grep conv1 | sed "s/^/$m /"
done | tee $O/time.txt
for i in 1 2 3; do
  for m in packed plain; do
    if [ $m = packed ]; then export DIRTORCH_AMD_PERSIST_PACK=1; else unset DIRTORCH_AMD_PERSIST_PACK; fi
    timeout 600 python bench.py --steps 30 --warmup 3 --cpu-seconds 0 --no-precision > $O/bench_${m}_$i.json 2> $O/err_${m}_$i.txt
  done
done
unset DIRTORCH_AMD_PERSIST_PACK
python - <<P
import json
for m in ('packed','plain'):
    v=[]
    for i in (1,2,3):
        try: v.append(json.loads(open('$O/bench_%s_%d.json'%(m,i)).read().strip().splitlines()[-1])['value'])
        except Exception as e: v.append(str(e)[:60])
    print(m, v)
P
