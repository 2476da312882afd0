#!/bin/bash
# round 5: does the autotuner (every admissible variant timed per layer shape) still find nothing over the heuristic?
O=gpurun_out/${1:-r5tune}; mkdir -p $O
timeout 900 python bench.py --autotune --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --layers > $O/bench_tuned.json 2> $O/layers_tuned.txt
timeout 300 python bench.py --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --layers > $O/bench_heur.json 2> $O/layers_heur.txt
for f in $O/bench_*.json; do echo $f $(python -c "import json;d=json.loads(open('$f').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"); done
python - <<PY
import re
def load(f):
    d={}
    for l in open(f):
        p=l.split()
        if len(p)>3 and p[3]=='ms': d[p[0]]=(p[1],float(p[2]))
    return d
a=load('$O/layers_heur.txt'); b=load('$O/layers_tuned.txt')
for k in a:
    if k in b and (a[k][0]!=b[k][0]): print('%-22s %-40s %.3f -> %-40s %.3f'%(k,a[k][0],a[k][1],b[k][0],b[k][1]))
PY
