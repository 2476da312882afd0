#!/bin/bash
# round 6, late: the whole -m gpu suite with the strided patch kernel (layer2.0), the packed weight stages of conv_patchw and the deep-X
# ring on layer3's conv1, then one-box step A/Bs of each against its switch.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6late}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
run() { timeout 600 python bench.py --steps 30 --warmup 3 --cpu-seconds 0 --no-precision > $O/bench_$1.json 2> $O/err_$1.txt; }
for i in 1 2; do
  run all_$i
  DIRTORCH_AMD_NO_PATCHW_PACK=1 run nopack_$i
  DIRTORCH_AMD_NO_PATCHS2=1 run nos2_$i
  DIRTORCH_AMD_X3_K2048=1 run nox3_$i
done
python - <<P
import json
for m in ('all','nopack','nos2','nox3'):
    v=[]
    for i in (1,2):
        try: v.append(json.loads(open('$O/bench_%s_%d.json'%(m,i)).read().strip().splitlines()[-1])['value'])
        except Exception as e: v.append(str(e)[:60])
    print(m, v)
P
