#!/bin/bash
# round 6: same-box A/B of the small-map picker rules (DIRTORCH_AMD_NO_SMALLMAP=1 = without them): batch 1 / 4, config A, and batch 1 on 1-6 streams
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6smallab}; mkdir -p $O
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
for rep in 1 2; do
for cfg in "--batch 1" "--batch 4" "--arch resnet50 --size 224 --batch 64"; do
  echo -n "new  [$cfg] "; timeout 300 python bench.py $cfg --steps 200 --warmup 10 --cpu-seconds 0 --no-precision 2>/dev/null | python -c "$pick"
  echo -n "old  [$cfg] "; DIRTORCH_AMD_NO_SMALLMAP=1 timeout 300 python bench.py $cfg --steps 200 --warmup 10 --cpu-seconds 0 --no-precision 2>/dev/null | python -c "$pick"
done; done 2>&1 | tee $O/ab.txt
timeout 300 python scripts/bench_batch1.py > $O/batch1_new.json 2>/dev/null; DIRTORCH_AMD_NO_SMALLMAP=1 timeout 300 python scripts/bench_batch1.py > $O/batch1_old.json 2>/dev/null
python - <<P
import json
for t in ('new','old'):
    d=json.load(open('$O/batch1_%s.json'%t)); print(t, {k:v['images_per_sec'] for k,v in d.items() if isinstance(v,dict)})
P
