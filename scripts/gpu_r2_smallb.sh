#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2l}
mkdir -p $O
pick='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms/step")'
for b in 1 2 4; do
  for mode in 0 auto; do
    echo -n "B=$b C3C1=$mode: "; DIRTORCH_AMD_C3C1=$mode timeout 600 python bench.py --cpu-seconds 0 --batch $b --steps 100 --warmup 5 --profile-every 10 --layers 2> $O/layers_b${b}_$mode.txt | tail -1 | python -c "$pick"
  done
done
grep -a "layer1" $O/layers_b1_0.txt $O/layers_b1_auto.txt | cut -c1-140
