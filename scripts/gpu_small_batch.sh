#!/bin/bash
pick='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "ms/step", d["ms_per_step"])'
for b in 1 4 32; do
  echo -n "B=$b heuristic: "; timeout 600 python bench.py --cpu-seconds 0 --batch $b --steps $((b<8?100:20)) --warmup 5 --profile-every 1000 2>/dev/null | tail -1 | python -c "$pick"
  echo -n "B=$b tuned:     "; timeout 600 python bench.py --cpu-seconds 0 --batch $b --steps $((b<8?100:20)) --warmup 5 --profile-every 1000 --autotune 2>/dev/null | tail -1 | python -c "$pick"
done
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_heads_gpu.py -q -m gpu -x 2>&1 | tail -2
