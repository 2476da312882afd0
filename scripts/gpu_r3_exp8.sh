#!/bin/bash
# Round-3 experiment 8: batch-size sweep of the headline workload (tile quantisation / Infinity Cache residency).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3k
mkdir -p $O
for b in 16 24 32 48 60; do
  python bench.py --cpu-seconds 0 --steps 20 --warmup 4 --batch $b > $O/b$b.json 2>/dev/null
  python - <<P
import json
d=json.load(open('$O/b$b.json')); rows={(r[0],r[1]):r[3] for r in d['roofline']['kernels']['rows']}
print('batch', $b, d['value'], 'img/s', d['ms_per_step'], 'ms/step', 'l3c1', rows.get(('256x256_persist1x1','layer3.conv1')), 'l3c2', rows.get(('512x128_patch3x3w','layer3.conv2')), 'l3c3', rows.get(('64x512_wreg1x1','layer3.conv3')))
P
done
