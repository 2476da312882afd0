#!/bin/bash
# A/B on one box: built-in tile heuristic vs a fresh autotune of every layer shape (table kept under gpurun_out/)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2t}
mkdir -p $O
pick='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms/step")'
echo -n "heuristic: "; timeout 300 python bench.py --cpu-seconds 0 | tail -1 | python -c "$pick"
echo -n "autotuned: "; DIRTORCH_AMD_TUNE_CACHE=$O/tuned.txt timeout 600 python bench.py --cpu-seconds 0 --autotune --layers 2> $O/layers_tuned.txt | tail -1 | python -c "$pick"
echo -n "heuristic: "; timeout 300 python bench.py --cpu-seconds 0 --layers 2> $O/layers_heur.txt | tail -1 | python -c "$pick"
echo -n "autotuned (replayed): "; DIRTORCH_AMD_TUNE_CACHE=$O/tuned.txt timeout 600 python bench.py --cpu-seconds 0 --autotune | tail -1 | python -c "$pick"
python - <<PY
def load(f):
    d={}
    for l in open(f):
        p=l.split()
        if len(p)>=4 and p[2].replace('.','').isdigit(): d[p[0]]=(p[1],float(p[2]))
    return d
a=load('$O/layers_heur.txt'); b=load('$O/layers_tuned.txt')
for k in a:
    if k in b and a[k][0]!=b[k][0]: print('%-22s %-38s %.3f -> %-38s %.3f'%(k,a[k][0],a[k][1],b[k][0],b[k][1]))
PY
