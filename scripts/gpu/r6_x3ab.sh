#!/bin/bash
# round 6: same-box A/B of the deep-X 1x1 ring on layer3's 1024 -> 256 conv1 (new picker rule) against the old K >= 2048 rule.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6x3ab}; mkdir -p $O
for i in 1 2 3; do
  for m in new old; do
    if [ $m = old ]; then export DIRTORCH_AMD_X3_K2048=1; else unset DIRTORCH_AMD_X3_K2048; fi
    timeout 600 python bench.py --steps 30 --warmup 3 --cpu-seconds 0 --no-precision > $O/bench_${m}_$i.json 2> $O/err_${m}_$i.txt
  done
done
unset DIRTORCH_AMD_X3_K2048
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python - <<P
import json
for m in ('new','old'):
    v=[json.loads(open('$O/bench_%s_%d.json'%(m,i)).read().strip().splitlines()[-1])['value'] for i in (1,2,3)]
    print(m, v)
P
