#!/bin/bash
# round 6: batch 1 at native sizes - picker rules for the tile counts below the 192-tile cliff (split-K there today), 1 / 2 / 4 streams
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6b1rules}; mkdir -p $O
for rep in 1 2; do
for m in 0 1 2 3; do
  DIRTORCH_AMD_SMALL_RULE=$m B1_SIZES=1024x1024,768x1024,683x1024,1024x819,500x375 B1_STREAMS=1,2,4 timeout 600 python scripts/bench_batch1.py > $O/m${m}_$rep.json 2> $O/m${m}_$rep.err
done
done
python - <<P
import json
for m in range(4):
    for rep in (1,2):
        d=json.loads(open('$O/m%d_%d.json'%(m,rep)).read().strip().splitlines()[-1])
        print('mode',m,'rep',rep,' '.join('%s:%s'%(k.replace('_streams','/s'),v['images_per_sec']) for k,v in d.items() if isinstance(v,dict)))
P
