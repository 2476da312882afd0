#!/bin/bash
# A/B of ONE environment switch on one box: bench.py with and without "$1" (e.g. DIRTORCH_AMD_NO_PATCHW_LC=1), twice each,
# interleaved; then the per-layer times of both legs for the layers matching $2 (default conv2).  $3 = output tag,
# BENCH_ARGS = extra bench.py flags (e.g. "--dtype fp16p").  How every kernel of rounds 2-4 was accepted or dropped.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
SW="$1"; PAT="${2:-conv2}"; O=gpurun_out/${3:-ab}
mkdir -p $O
pick='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms/step")'
for i in 1 2; do
  echo -n "base : "; timeout 300 python bench.py $BENCH_ARGS --cpu-seconds 0 --layers 2> $O/layers_base.txt | tail -1 | python -c "$pick"
  echo -n "$SW : "; env $SW timeout 300 python bench.py $BENCH_ARGS --cpu-seconds 0 --layers 2> $O/layers_sw.txt | tail -1 | python -c "$pick"
done
python - <<PY
import re
def load(f):
    d={}
    for l in open(f):
        p=l.split()
        if len(p)>=4 and re.match(r'^[0-9.]+$', p[2]): d[p[0]]=(p[1],float(p[2]))
    return d
a=load('$O/layers_base.txt'); b=load('$O/layers_sw.txt')
n=0
for k in list(a)+[k for k in b if k not in a]:
    if re.search(r'$PAT', k) and n < 20:
        x, y = a.get(k, ('-', 0.0)), b.get(k, ('-', 0.0))
        print('%-22s %-36s %.3f -> %-36s %.3f'%(k,x[0],x[1],y[0],y[1])); n+=1
PY
