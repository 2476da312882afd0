#!/bin/bash
# round 5: the Infinity-Cache probe + per-layer times of the step at sub-batches 8 / 16 / 32 on ONE box
# (does a layer3 whose tensors fit the 256 MiB cache run faster per image?)   gpurun -- 'bash scripts/gpu/r5_mall.sh <tag>'
TAG=${1:-r5mall}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/probes/mall_probe.hip -o /tmp/mall_probe && timeout 300 /tmp/mall_probe > $OUT/mall_probe.txt 2>&1
for B in 8 16 32; do
  timeout 300 python bench.py --dtype fp16p --batch $B --steps 30 --warmup 3 --profile-every 3 --cpu-seconds 0 --layers \
      > $OUT/bench_b$B.json 2> $OUT/layers_b$B.txt
done
tail -n 60 $OUT/mall_probe.txt
