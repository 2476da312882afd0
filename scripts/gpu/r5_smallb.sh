#!/bin/bash
# round 5: per-layer times of the whole step at batch 1 / 2 / 4 - do layer1 / layer2 run faster per image once their
# tensors (ping-pong maps + t1 + t2) fit the 256 MiB Infinity Cache?   gpurun -- 'bash scripts/gpu/r5_smallb.sh <tag>'
TAG=${1:-r5smallb}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for B in 1 2 4; do
  DIRTORCH_AMD_C3C1=force timeout 300 python bench.py --dtype fp16p --batch $B --steps 40 --warmup 3 --profile-every 4 --cpu-seconds 0 --layers \
      > $OUT/bench_b$B.json 2> $OUT/layers_b$B.txt
done
grep -E "layer1|layer2" $OUT/layers_b2.txt
