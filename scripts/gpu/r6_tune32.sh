#!/bin/bash
# round 6: the autotuner over the enlarged variant table at the headline configuration - does any table entry beat the picker's choice?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r6tune32}; mkdir -p $O
timeout 600 python bench.py --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --no-precision --layers > $O/bench_h.json 2> $O/layers_h.txt
timeout 900 python bench.py --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --no-precision --layers --autotune > $O/bench_t.json 2> $O/layers_t.txt
python - <<P
import json
for t in ('h','t'):
    d=json.loads(open('$O/bench_%s.json'%t).read().strip().splitlines()[-1]); print(t, d['value'], d['ms_per_step'])
def load(f):
    d={}
    for l in open(f):
        p=l.split()
        if len(p)>=4:
            try: d[p[0]]=(p[1],float(p[2]))
            except ValueError: pass
    return d
a=load('$O/layers_h.txt'); b=load('$O/layers_t.txt')
for k in a:
    if k in b and a[k][0]!=b[k][0]:
        print('%-20s %-34s %.3f -> %-34s %.3f'%(k,a[k][0],a[k][1],b[k][0],b[k][1]))
P
