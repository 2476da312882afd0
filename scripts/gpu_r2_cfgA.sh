#!/bin/bash
cd "$GRAFT_REPO_ROOT"
pick='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["ms_per_step"], "ms/step", d["roofline"]["step_mfma_frac"])'
for b in 64 128 256 512; do echo -n "R50 224 B=$b: "; timeout 300 python bench.py --cpu-seconds 0 --arch resnet50 --size 224 --batch $b --steps 50 --warmup 5 | tail -1 | python -c "$pick"; done
timeout 300 python bench.py --cpu-seconds 0 --arch resnet50 --size 224 --batch 256 --steps 30 --warmup 5 --layers 2>&1 >/dev/null | sort -k3 -n -r | head -14 | cut -c1-120
