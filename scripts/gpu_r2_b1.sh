#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2o}
mkdir -p $O
pick='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms/step")'
for i in 1 2; do
for b in 1 2 4 8 16; do echo -n "B=$b: "; timeout 600 python bench.py --cpu-seconds 0 --batch $b --steps $((b<8?200:40)) --warmup 10 --profile-every 100000 2>/dev/null | tail -1 | python -c "$pick"; done
done | tee $O/small_batch.txt
echo -n "B=1 all fused forms off: "; DIRTORCH_AMD_C3C1=0 DIRTORCH_AMD_NO_DUAL=1 DIRTORCH_AMD_NO_X3=1 DIRTORCH_AMD_NO_PATCHS=1 timeout 600 python bench.py --cpu-seconds 0 --batch 1 --steps 200 --warmup 10 --profile-every 100000 2>/dev/null | tail -1 | python -c "$pick"
timeout 300 python bench.py --cpu-seconds 0 --batch 1 --steps 50 --warmup 10 --profile-every 1 --layers 2> $O/layers_b1.txt | tail -1 | python -c "$pick"
sort -k3 -n -r $O/layers_b1.txt | head -25 | cut -c1-120
